for t in 1 2 4 8 1 4; do echo "taper $t"; PT_PATHS_TAPER=$t timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' || exit 1; done
