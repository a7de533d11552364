for rep in 1 2; do for lib in "$@"; do for wl in "c5 8" "c4 4"; do echo -n "$lib "; PT_LIB=$PWD/$lib timeout -k 10 120 python3 tools/mb/spp_scale.py $wl | cut -c1-60 || exit 1; done; done; done
