SDMI_NO_REGRESSION_GATE=1 tools/profile_round.sh r05 > gpurun_out/r05_profile.log 2>&1 || exit 1
cp gpurun_out/r05/p/r05_* profiles/
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1 || exit 2
tools/round_benches.sh r05 > gpurun_out/r05_benches.log 2>&1
echo done
