#!/bin/bash
# the general-mesh solver at the C3 size (160^3 wavy hexahedra as a polyhedral mesh, 10 M particles, pimple): bench line + kernel stats
cd /root/repo; export TMPDIR=/tmp
TAG=${TAG:-now}
python tools/ldu_bench.py 160 5 wavy 10000000 mg 1e-6 pimple 2>&1 | grep "^{" | tee gpurun_out/ldu_c3_$TAG.jsonl | cut -c1-420
KSTATS_TOP=${KSTATS_TOP:-45} bash tools/kstats.sh ldu_c3_$TAG -- python /root/repo/tools/ldu_bench.py 160 5 wavy 10000000 mg 1e-6 pimple | tee gpurun_out/ldu_c3_kernels_$TAG.txt
