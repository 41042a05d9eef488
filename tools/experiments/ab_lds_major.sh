#!/bin/bash
# GPU-box script: the local-majorant kernel with the majorant grid in LDS against the same grid read through L2
for w in config3_textbook config4_delta_4k_256spp_cloud1024; do for lv in ${LEVELS:-2 3}; do for off in 1 0; do
VR_TUNE_MAJORANT_LEVEL=$lv VR_TUNE_NO_LDS_MAJOR=$off timeout -k 10 200 python bench.py --workload $w --mode local --steps 3 --warmup 1 --cpu-baseline off --pipeline off 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w level=$lv no_lds=$off', round(d['value'],1), round(d['roofline']['kernel_ms_per_launch'],1), d['config']['frame_mean_rgb'][0])" || exit 1
done; done; done
