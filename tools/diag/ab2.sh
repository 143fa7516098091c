python bench.py --workload llama2_7b_semseg_B32_L1024_C12 --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-extra-configs 2>&1 | tail -1 | cut -c1-200
python bench.py --workload llama3_8b_recon_B32_L1024_C12 --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-extra-configs 2>&1 | tail -1 | cut -c1-200
