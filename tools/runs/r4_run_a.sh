#!/bin/bash
# round-4 GPU call A: telemetry probe, the new tests, a baseline bench line, and the cache-policy A/B of the resident GEMM
# (time + shader clock + hwmon power per arm; FETCH_SIZE / WRITE_SIZE / L2 hit per arm in separate --pmc passes).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
{
  echo "== hwmon"; ls /sys/class/drm/ 2>&1 | head; for f in /sys/class/drm/card*/device/hwmon/hwmon*/; do echo $f; ls $f | tr '\n' ' '; echo; done
  for f in /sys/class/drm/card*/device/hwmon/hwmon*/{power1_average,power1_input,power1_cap,freq1_input,freq2_input,temp1_input}; do [ -r $f ] && echo "$f = $(cat $f 2>&1)"; done
  echo "== rocm-smi"; timeout 30 rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40
  echo "== amd-smi"; timeout 30 amd-smi metric -p -c 2>&1 | head -60
} > $O/telemetry.txt 2>&1
tail -30 $O/telemetry.txt
timeout 900 python -m pytest tests/test_gpu_optim.py tests/test_gpu_ops.py -m gpu -q -x -k "optim or mirror or loss_scale or grad_stats or attention_bwd or persistent" > $O/pytest_new.log 2>&1; echo "pytest(new) rc=$?"; tail -4 $O/pytest_new.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4a/bench.json').read().strip().splitlines()[-1])
print('train ms',j['ms_per_step'],'roof',j['roofline']['frac'],'fwd',j['fwd']['ms_per_step'],j['fwd']['mfma_frac'],j['fwd']['roofline']['frac'])
PY
CASES="g3:50432:2304:768:0:8 g3:50432:768:768:2:8 g3:50432:3072:768:1:8 g3:50432:3072:768:7:8 g3:50432:768:3072:2:8 g3:50432:3072:768:6:8 g3:50432:768:3072:0:8 g3:50432:768:2304:0:8"
for round in 1 2; do
for V in "" _cnt _csc1 _csc1rnt _csc1ant _cntant; do
  B=$R/tools/_build$V
  [ -x $B/gemm_dev ] || continue
  echo "== arm ${V:-_base} (pass $round)"
  PW=""; [ $round = 1 ] && PW="--power 1.0"
  timeout 300 $B/gemm_dev --check --iters 30 $PW $CASES > $O/gd${V:-_base}_$round.txt 2>&1
  grep -E "TF/s|shader clock|power:" $O/gd${V:-_base}_$round.txt | grep -v "^  wave\|item"
done
done
cd /tmp && export TMPDIR=/tmp
PCASES="g3:50432:2304:768:0 g3:50432:768:768:2 g3:50432:3072:768:1 g3:50432:3072:768:7 g3:50432:768:3072:2 g3:50432:3072:768:6"
for V in "" _csc1 _csc1ant _cntant; do
  B=$R/tools/_build$V
  [ -x $B/gemm_dev ] || continue
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc${V:-_base}/p$i -o p -- $B/gemm_dev --iters 3 $PCASES > $O/pmc${V:-_base}_p$i.log 2>&1
  done
  find $O/pmc${V:-_base} -name "*agent*" -delete
  python $R/tools/r4_pmc_fold.py $O/pmc${V:-_base}
done
du -sh $O
