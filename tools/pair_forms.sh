# builds the lane-pair decoder's slot write in the forms of roc_lane.h's VIDC_PAIR_OLD_FORM (investigation of the round-3 corruption)
cd /root/repo; mkdir -p tools/_bin
for f in ${FORMS:-1 2 3 4 5 6}; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -DVIDC_PAIR_OLD_FORM=$f -I include -o tools/_bin/libvidc_form$f.so vector_db_id_compression_amd/csrc/{ctx,ef,packed,roc,wt}.hip 2>&1 | tail -3 ) &
done
wait; ls -la tools/_bin/libvidc_form*.so
