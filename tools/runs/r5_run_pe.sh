#!/bin/bash
# round 5: patch embed gathered inside the GEMM (item 9): parity tests + A/B timing
mkdir -p gpurun_out/r5pe
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -k "patch_embed" > gpurun_out/r5pe/pytest_pe.txt 2>&1
tail -15 gpurun_out/r5pe/pytest_pe.txt
timeout 300 python -m pytest tests/test_gpu_encoder.py -x -q -k "tokenizer or multimodal" > gpurun_out/r5pe/pytest_tok.txt 2>&1
tail -5 gpurun_out/r5pe/pytest_tok.txt
timeout 300 python tools/patch_embed_time.py > gpurun_out/r5pe/patch_embed_time.txt 2>&1
cat gpurun_out/r5pe/patch_embed_time.txt
