#!/bin/bash
O=gpurun_out/r02_attn; mkdir -p $O
bash tools/sessions/r02_attn.sh | grep -v "^{" || exit 1
timeout -k 10 300 python3 tools/s3_forward_probe.py > $O/probe.json 2>>$O/err.log; cat $O/probe.json
