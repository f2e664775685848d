#!/usr/bin/env python3
"""Join rocprofv3 output with the engine's launch log (SDMI_LAUNCH_LOG): which SHAPE each kernel of a denoising step ran.

  kernel trace  : join_trace.py time <kernel_trace.csv> <launch_log.txt> [step] [--json step_families.json]
                  -> per shape: launches, us, TF/s, weight GB/s (and the family sums as JSON: what bench.py quotes as roofline.rocprofv3)
  PMC passes    : join_trace.py pmc <fetch counter_collection.csv> <write counter_collection.csv> <launch_log.txt> <out.json>
                  -> per shape: FETCH (x2, gfx950 correction, MI355X_MICROARCH.md) + WRITE bytes per step vs the
                     algorithmic bytes (weights once + activations in/out once)
  MFMA busy     : join_trace.py mfma <counter_collection.csv with SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE> <launch_log.txt> <out.json> [step_families.json]
                  (with the families file: the busy cycles also over the family's kernel time of the TRACE pass x 2.4 GHz)

  regression    : join_trace.py compare <old step_by_shape.txt> <new step_by_shape.txt> [--tol 0.05] [--allow reasons.txt] [--no-drift]
                  -> per LOGICAL shape (tile config and split-K factor dropped: a re-tuned plan is the same work), the step's
                     time for it in both profiles; exit code 1 when a shape got more than --tol slower (and more than 1 us)
                     unless the allow file lists it ("<shape substring> :: <reason>" per line)

A step = the kernels from stem_conv to cfg_ddpm (or final_conv_step, which contains it); log line i is matched to the i-th kernel of the step whose name fits the
line's kind (fill kernels of hipMemsetAsync and anything else unknown are skipped)."""
import collections
import csv
import json
import re
import sys

KIND_PAT = {"igemm": "igemm_kernel", "halo": "conv3_halo_kernel", "finalize": "splitk_finalize", "attn": "attn_kernel",
            "gn_fused": "gn_fused_kernel", "gn_fused_slab": "gn_fused_kernel", "gn_stats": "gn_stats_kernel", "gn_apply": "gn_apply_kernel",
            "gn_apply_acc": "gn_apply_kernel",
            "layernorm": "layernorm_kernel", "stem": "stem_conv", "final_conv": "final_conv", "xattn": "xattn", "b2b": "b2b_kernel"}


def read_rows(path):
    """kernel rows of a rocprofv3 run: a kernel_trace CSV, or the rocpd sqlite database newer rocprofv3 writes by default"""
    if path.endswith(".db"):
        import sqlite3
        con = sqlite3.connect(path)
        cur = con.execute("select name, start, end, grid_x, workgroup_x from kernels")
        return [{"Kernel_Name": n, "Start_Timestamp": str(s), "End_Timestamp": str(e), "Grid_Size_X": g, "Workgroup_Size_X": w}
                for n, s, e, g, w in cur]
    return list(csv.DictReader(open(path)))


def read_log(path):
    out = []
    for ln in open(path):
        ln = ln.strip()
        if not ln:
            continue
        kind = ln.split()[0]
        kv = dict(re.findall(r"(\w+)=(\S+)", ln))
        out.append((kind, kv, ln))
    return out


def steps_of(rows):
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "stem_conv" in r["Kernel_Name"]]
    ends = [i for i, r in enumerate(rows) if "cfg_ddpm" in r["Kernel_Name"] or "final_conv_step" in r["Kernel_Name"]]
    out = []
    for s in starts:
        e = next((x for x in ends if x > s), None)
        if e is not None:
            out.append(rows[s:e + 1])
    return out


def align(step_rows, log):
    """-> list of (log entry, row) ; raises if a log entry finds no kernel"""
    j, out = 0, []
    for kind, kv, text in log:
        pat = KIND_PAT.get(kind, kind)
        while j < len(step_rows) and pat not in step_rows[j]["Kernel_Name"]:
            j += 1
        if j >= len(step_rows):
            raise SystemExit(f"launch log entry '{text}' has no kernel left in the step ({len(step_rows)} kernels)")
        out.append(((kind, kv, text), step_rows[j]))
        j += 1
    return out


def shape_key(kind, kv):
    if kind in ("igemm", "halo", "finalize"):
        return f"{kind:8s} M={kv['M']:>5s} N={kv['N']:>5s} K={kv['K']:>5s}" + (f" ks={kv['ks']} s={kv['s']} up={kv['up']} {kv['cfg']} split {kv['split']}" if kind != "finalize" else f" split {kv['split']}")
    if kind == "attn":
        return f"attn     d={kv['d']} Sq={kv['Sq']} Skv={kv['Skv']}"
    if kind.startswith("gn") or kind in ("layernorm", "b2b"):
        return f"{kind:8s} " + " ".join(f"{k}={v}" for k, v in kv.items())
    return kind


def family_of(kind):
    if kind in ("igemm", "halo", "b2b"):
        return "mfma"
    if kind == "finalize":
        return "finalize"
    if kind in ("attn", "xattn"):
        return "attn"
    if kind.startswith(("gn", "layer")):
        return "norm"
    return "other"


def cmd_time(trace, logp, which=None, json_out=None):
    rows = read_rows(trace)
    steps = steps_of(rows)
    log = read_log(logp)
    step = steps[int(which) if which is not None else min(15, len(steps) - 1)]
    agg = collections.OrderedDict()
    for (kind, kv, _), r in align(step, log):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(shape_key(kind, kv), [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += us
        a[2] += float(kv.get("flops", 0))
        a[3] += float(kv.get("wbytes", 0))
    tot = sum(a[1] for a in agg.values())
    print(f"{len(step)} kernels in the step, {sum(a[0] for a in agg.values())} matched, {tot:.1f} us")
    fams = collections.OrderedDict()
    for (kind, kv, _), r in align(step, log):
        f = fams.setdefault(family_of(kind), [0, 0.0, 0.0])
        f[0] += 1
        f[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        f[2] += float(kv.get("flops", 0))
    for f, (n, us, fl) in fams.items():
        print(f"  family {f:9s} x{n:3d} {us:8.1f} us" + (f"  {fl / us * 1e-6:7.1f} TF/s" if fl else ""))
    if json_out:
        j = {"note": "one denoising step of bench.py under rocprofv3 --kernel-trace, kernels joined with the engine's launch log "
                     "(tools/join_trace.py time); ms = sum of the family's kernel durations, tflops = executed FLOPs / that",
             "kernels_per_step": len(step), "step_kernel_ms": round(tot / 1e3, 4)}
        for f, (n, us, fl) in fams.items():
            j[f] = {"launches": n, "ms": round(us / 1e3, 4), "tflops": round(fl / us * 1e-6, 2) if fl else None}
        json.dump(j, open(json_out, "w"), indent=1)
    for k, (n, us, fl, wb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        extra = f"  {fl / us * 1e-6:7.1f} TF/s  {wb / us * 1e-3:7.1f} GB/s weights" if fl else ""
        print(f"  {k:78s} x{n:3d} {us:8.1f} us  avg {us / n:7.2f}{extra}")


def pmc_rows(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    return steps_of(rows)


def cmd_pmc(fetch_csv, write_csv, logp, outp):
    log = read_log(logp)
    res = collections.OrderedDict()
    nsteps = {}
    for counter, path, scale in (("FETCH_SIZE", fetch_csv, 2 * 1024.0), ("WRITE_SIZE", write_csv, 1024.0)):
        steps = pmc_rows(path, counter)[-8:]
        nsteps[counter] = len(steps)
        for step in steps:
            for (kind, kv, _), r in align(step, log):
                e = res.setdefault(shape_key(kind, kv), {"launches_per_step": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "algorithmic": 0.0, "kind": kind})
                e[counter] += float(r["Counter_Value"]) * scale / len(steps)
                if counter == "FETCH_SIZE":
                    e["launches_per_step"] += 1.0 / len(steps)
                    if kind in ("igemm", "halo"):
                        M, N, K = int(kv["M"]), int(kv["N"]), int(kv["K"])
                        ks = int(kv["ks"])
                        a_in = M * (K // (ks * ks)) * 2 * (int(kv["s"]) ** 2) / (4 if kv["up"] == "1" else 1)
                        out_b = M * N * (6 if kv.get("out32") == "1" else 2) + (M * N * 4 if kv.get("res") == "1" else 0)
                        e["algorithmic"] += (N * K * 2 + a_in + out_b) / len(steps)
    fam = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for k, e in res.items():
        # ONE family definition, shared with bench.py and cmd_mfma: "mfma" = the GEMM kernels that run MFMAs (implicit-GEMM conv /
        # linear, halo conv, back-to-back GEMM); "finalize" = the combine launches of split-K GEMMs, reported on their own
        f = family_of(e["kind"])
        fam[f][0] += e["FETCH_SIZE"]; fam[f][1] += e["WRITE_SIZE"]; fam[f][2] += e["launches_per_step"]
        e["hbm_bytes"] = e["FETCH_SIZE"] + e["WRITE_SIZE"]
        e["launches_per_step"] = round(e["launches_per_step"], 2)
    out = {"note": "per denoising step, mean of the last 8 steps of each pass; FETCH_SIZE x2 (gfx950 reports half of wide reads, "
                   "MI355X_MICROARCH.md); separate rocprofv3 --pmc passes; joined with the engine's launch log by tools/join_trace.py. "
                   "'algorithmic' = weights once + input activations once + outputs (and residual) once, per GEMM shape.",
           "steps_averaged": nsteps, "launch_log_entries": len(log),
           "families": {f: {"fetch_bytes": v[0], "write_bytes": v[1], "hbm_bytes": v[0] + v[1], "launches": round(v[2])} for f, v in fam.items()},
           "whole_step": {"hbm_bytes": sum(v[0] + v[1] for v in fam.values()), "launches": round(sum(v[2] for v in fam.values()))},
           "by_shape": collections.OrderedDict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes"]))}
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps(out["families"], indent=1))
    for k, e in list(out["by_shape"].items())[:25]:
        print(f"  {k:78s} x{e['launches_per_step']:5.1f} fetch {e['FETCH_SIZE']/1e6:8.1f} MB write {e['WRITE_SIZE']/1e6:7.1f} MB  algorithmic {e['algorithmic']/1e6:7.1f} MB")


def cmd_mfma(csv_path, logp, outp, fam_json=None):
    log = read_log(logp)
    fam_ms = json.load(open(fam_json)) if fam_json else {}
    allrows = list(csv.DictReader(open(csv_path)))
    by_counter = collections.defaultdict(list)
    for r in allrows:
        by_counter[r["Counter_Name"]].append(r)
    per = collections.OrderedDict()
    for counter, rows in by_counter.items():
        steps = steps_of(rows)[-8:]
        for step in steps:
            for (kind, kv, _), r in align(step, log):
                fam = family_of(kind)
                e = per.setdefault(fam, collections.defaultdict(float))
                e[counter] += float(r["Counter_Value"]) / len(steps)
                if counter == "SQ_VALU_MFMA_BUSY_CYCLES":
                    e["launches"] += 1.0 / len(steps)
    out = {"note": "per denoising step; SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES summed over the launches of each family "
                   "(rocprofv3 --pmc, one pass; both counters are summed over all SEs/XCDs by rocprofv3). "
                   "mfma_busy_frac = MFMA-busy cycles / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs) when GRBM_GUI_ACTIVE is present, "
                   "else / SQ_BUSY_CYCLES-normalised as stated per entry.", "families": {}}
    for fam, e in per.items():
        d = dict(e)
        if d.get("SQ_BUSY_CYCLES"):
            d["mfma_busy_over_sq_busy"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / d["SQ_BUSY_CYCLES"]
        if d.get("GRBM_GUI_ACTIVE"):
            d["mfma_busy_frac_of_chip"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            d["clock_basis"] = ("GRBM_GUI_ACTIVE of THIS counter pass (rocprofv3 serialises the kernels under --pmc): busy cycles of the "
                                "profiled pass, not wall-clock cycles of the un-profiled step")
        if isinstance(fam_ms.get(fam), dict) and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            wall_cycles = fam_ms[fam]["ms"] * 1e-3 * 2.4e9
            d["mfma_busy_frac_of_family_wall_time"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (wall_cycles * 1024.0), 4)
            d["wall_time_basis"] = f"{fam_ms[fam]['ms']} ms of kernel time in the kernel-trace pass x 2.4 GHz nominal x 1024 SIMDs"
        out["families"][fam] = d
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps(out["families"], indent=1))


SHAPE_LINE = re.compile(r"^\s{2}(?!family)(.+?)\s+x\s*(\d+)\s+([0-9.]+) us\s+avg\s+([0-9.]+)")


def logical_shape(key):
    """the work a line stands for, whatever tile plan ran it: 'igemm M= 8192 N= 320 K= 3520 ks=3 s=1 up=0 t128x128s3q2 split 1' and
    the halo kernel's line for the same conv both become 'conv M=8192 N=320 K=3520 ks=3 s=1 up=0'; the combine of a split-K GEMM
    is folded into its GEMM (a plan may trade one for the other)"""
    key = " ".join(key.split())
    m = re.match(r"(igemm|halo) M= ?(\d+) N= ?(\d+) K= ?(\d+) ks=(\d) s=(\d) up=(\d)", key)
    if m:
        return f"conv M={m.group(2)} N={m.group(3)} K={m.group(4)} ks={m.group(5)} s={m.group(6)} up={m.group(7)}"
    m = re.match(r"finalize M= ?(\d+) N= ?(\d+) K= ?(\d+)", key)
    if m:
        return f"finalize M={m.group(1)} N={m.group(2)} K={m.group(3)}"
    key = re.sub(r" flops=\S+", "", key)
    key = re.sub(r" split=\d+", "", key)          # gn_fused_slab follows its conv's split-K factor
    return key


def read_shapes(path):
    out = collections.OrderedDict()
    for ln in open(path):
        m = SHAPE_LINE.match(ln)
        if not m:
            continue
        k = logical_shape(m.group(1))
        e = out.setdefault(k, [0, 0.0])
        e[0] += int(m.group(2))
        e[1] += float(m.group(3))
    # a split-K conv and its combine are one unit of work: fold "finalize M N K" into "conv M N K ..." when exactly one conv has
    # that (M, N, K); several convs share a combine shape only when their ks / stride differ, then the combine stays its own line
    for k in [k for k in out if k.startswith("finalize ")]:
        mnk = k[len("finalize "):]
        hosts = [c for c in out if c.startswith("conv " + mnk + " ")]
        if len(hosts) == 1:
            out[hosts[0]][1] += out[k][1]
            del out[k]
    return out


def cmd_compare(oldp, newp, tol=0.05, allow=None, floor_us=1.0, use_drift=True):
    old, new = read_shapes(oldp), read_shapes(newp)
    reasons = []
    if allow:
        for ln in open(allow):
            ln = ln.strip()
            if ln and not ln.startswith("#") and "::" in ln:
                pat, why = ln.split("::", 1)
                reasons.append((pat.strip(), why.strip()))
    t_old, t_new = sum(v[1] for v in old.values()), sum(v[1] for v in new.values())
    print(f"step kernel time {t_old:.1f} -> {t_new:.1f} us ({(t_new / t_old - 1) * 100:+.1f} %), {len(old)} -> {len(new)} logical shapes")
    # The two profiles come from two boxes of a pool whose steps differ by 2 - 3 % for one binary (power-limited clocks): a shape is
    # judged against the DRIFT of the whole comparison = the median per-launch ratio over the shapes both profiles hold, clamped to
    # +-3 % (a uniform slowdown beyond that is a regression of the step, which the total above and the whole-step check report).
    ratios = sorted((new[k][1] / max(new[k][0], 1)) / (old[k][1] / max(old[k][0], 1)) for k in set(old) & set(new) if old[k][1] > 0)
    drift = ratios[len(ratios) // 2] if (use_drift and ratios) else 1.0
    drift = min(max(drift, 0.97), 1.03)
    print(f"box drift (median per-launch ratio of {len(ratios)} common shapes, clamped to +-3 %): {drift:.3f}")
    bad = []
    rows = []
    for k in sorted(set(old) | set(new), key=lambda k: -(new.get(k, [0, 0.0])[1] - old.get(k, [0, 0.0])[1])):
        o, n = old.get(k), new.get(k)
        if o is None:
            rows.append(f"  NEW      {k:70s} x{n[0]:3d} {n[1]:8.1f} us")
            continue
        if n is None:
            rows.append(f"  GONE     {k:70s} x{o[0]:3d} {o[1]:8.1f} us")
            continue
        d = n[1] - o[1]
        # slower per launch by more than tol beyond the drift AND by more than floor_us per launch (launches of 5 - 20 us jitter by
        # 0.5 - 1 us from box to box and run to run: five profiles of one binary, round 5)
        per_o, per_n = o[1] / max(o[0], 1), n[1] / max(n[0], 1)
        slow = per_n > (1 + tol) * drift * per_o and per_n - drift * per_o > floor_us
        why = next((w for p, w in reasons if p in k), None) if slow else None
        tag = "ok"
        if slow:
            tag = "ALLOWED" if why else "SLOWER"
            if not why:
                bad.append(k)
        if slow or abs(d) > floor_us:
            rows.append(f"  {tag:8s} {k:70s} x{o[0]:3d}->{n[0]:3d} {o[1]:8.1f} -> {n[1]:8.1f} us ({d:+7.1f})" + (f"   [{why}]" if why else ""))
    print("\n".join(rows))
    # shapes that vanished or appeared change the step only through the total: the gate on them is the step time itself
    if t_new > (1 + tol) * t_old:
        bad.append(f"whole step {t_old:.1f} -> {t_new:.1f} us")
    if bad:
        print(f"REGRESSION: {len(bad)} shape(s) more than {tol * 100:.0f} % slower per launch than {oldp} with no reason on file:")
        for k in bad:
            print("   ", k)
        return 1
    print("no unexplained per-shape regression")
    return 0


if __name__ == "__main__":
    c = sys.argv[1]
    if c == "compare":
        rest = sys.argv[2:]
        tol, allow = 0.05, None
        if "--tol" in rest:
            i = rest.index("--tol"); tol = float(rest[i + 1]); del rest[i:i + 2]
        if "--allow" in rest:
            i = rest.index("--allow"); allow = rest[i + 1]; del rest[i:i + 2]
        use_drift = "--no-drift" not in rest
        if not use_drift:
            rest.remove("--no-drift")
        raise SystemExit(cmd_compare(rest[0], rest[1], tol, allow, use_drift=use_drift))
    if c == "time":
        rest = sys.argv[2:]
        jout = None
        if "--json" in rest:
            i = rest.index("--json")
            jout = rest[i + 1]
            del rest[i:i + 2]
        cmd_time(*rest[:3], json_out=jout)
    elif c == "pmc":
        cmd_pmc(*sys.argv[2:6])
    elif c == "mfma":
        cmd_mfma(*sys.argv[2:6])
    else:
        raise SystemExit(__doc__)
