run() { echo "$*"; env "$@" timeout -k 10 120 python3 tools/mb/spp_scale.py $WL || exit 1; }
for WL in "c5 8" "c4 4"; do
run PT_X=0
run PT_LIB=$PWD/pathtracer_amd/libpt_w4.so
run PT_LIB=$PWD/pathtracer_amd/libpt_w6.so
run PT_PATHS_BLOCK=512
run PT_STREAM_TILE=128
run PT_X=0
done
