#!/bin/bash
# round-4 GPU call X: side-stream test, bench line with the schedule object, other workloads with / without the overlap
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4x
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -k "side_stream or fused_gradient or training_steps" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4x/bench.json').read().strip().splitlines()[-1])
print('train', j['ms_per_step'], j['value'], 'schedule', j['schedule']['wgrad_side_stream'], j['schedule']['serial_ms_per_step'], 'roof', j['roofline']['frac'], j['roofline']['avg_launch_us'], 'wgrad', j['roofline']['wgrad_kernel']['avg_launch_us'], 'fwd', j['fwd']['ms_per_step'])
PY
for W in large512 large1568 mixed; do
  for A in 1 0; do
    ME_WGRAD_OVERLAP=$A timeout 300 python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-fwd-leg 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W overlap=$A', j['ms_per_step'], j['value'])"
  done
done 2>&1 | tee $O/other.txt
