#!/bin/bash
cd /root/repo
timeout 900 python tools/ldu_bench.py 160 5 wavy 10000000 mg 1e-6 pimple 2>&1 | tail -1 | tee gpurun_out/ldu_c3_like.json
timeout 900 python tools/ldu_bench.py 160 5 lattice 10000000 mg 1e-6 pimple 2>&1 | tail -1 | tee -a gpurun_out/ldu_c3_like.json
