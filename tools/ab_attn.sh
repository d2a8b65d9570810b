#!/bin/bash
# same-box A/B: bench with the current library, then with tools/_build_base/libmetaenc_prevattn.so (previous attention kernels) -- built by tools/build_prevattn.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/ab
cp metatransformer_amd/libmetaenc.so /tmp/cur.so
for V in cur prev cur prev; do
  [ $V = cur ] && cp /tmp/cur.so metatransformer_amd/libmetaenc.so || cp tools/_build_base/libmetaenc_prevattn.so metatransformer_amd/libmetaenc.so
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab/$V.json 2> gpurun_out/ab/$V.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab/$V.json").read().strip().splitlines()[-1])
o=j["other_kernels"]
print("$V", "train ms",j["ms_per_step"],"roof",j["roofline"]["frac"],"fwd",j["fwd"]["ms_per_step"],j["fwd"]["roofline"]["frac"], "attn fwd/bwd", o["attention_fwd"]["avg_launch_us"], o["attention_bwd"]["avg_launch_us"])
PY
done
cp /tmp/cur.so metatransformer_amd/libmetaenc.so
