#!/bin/bash
for m in f32 bf16x3; do D=48 B=32 L=14 SEED=17 CLIORA_MFMA=$m python tools/accuracy.py 2>/dev/null | tail -1; done
for m in bf16x3; do D=64 B=32 L=14 SEED=17 CLIORA_MFMA=$m python tools/accuracy.py 2>/dev/null | tail -1; done
