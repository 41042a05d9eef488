#!/bin/bash
# GPU-box script: triangles per BVH leaf (VR_TUNE_BVH_LEAF) on the mesh workloads
for w in ${WLS:-config5_tracker_ratio_sphere_4k config5_tracker_delta_sphere_4k config5_marcher_nee_sphere_4k config5_marcher_nee_cornell_4k}; do for leaf in ${LEAVES:-1 2 4}; do
VR_TUNE_BVH_LEAF=$leaf timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 1 --cpu-baseline off --pipeline off 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w leaf=$leaf', round(d['value'],1), round(d['roofline']['kernel_ms_per_launch'],2))" || exit 1
done; done
