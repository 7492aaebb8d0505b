#!/bin/bash
for f in 65536 1048576; do FOLD=$f timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 2>&1 | grep "fold\|batched="; done | tee gpurun_out/r3q_time_fold.log
