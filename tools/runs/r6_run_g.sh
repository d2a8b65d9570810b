#!/bin/bash
# round 6, call G: tools/energy_probe (marginal pJ / flop and pJ / byte of MFMA, LDS, L2, HBM from socket power -> profiles/r06_energy_probe.txt); the
# spectrogram tokenizer's stride-10 patches through the fused patch-embed kernels (tests)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6g
mkdir -p $O
cd $R
timeout 300 tools/_build/energy_probe > $O/energy_probe.txt 2>&1; echo "energy rc=$?"; cat $O/energy_probe.txt
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "patch_embed" > $O/tests_pe.txt 2>&1; echo "pe rc=$?"; tail -5 $O/tests_pe.txt
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -k "acoustic or tokenizer" > $O/tests_tok.txt 2>&1; echo "tok rc=$?"; tail -3 $O/tests_tok.txt
timeout 300 python tools/patch_embed_time.py > $O/pe_time.txt 2>&1; echo "time rc=$?"; tail -12 $O/pe_time.txt
