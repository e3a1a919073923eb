# A/B of two builds of the library in one GPU session (same box, alternating)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RUN_TAG:-ab}
mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in ${LIBS:-libbeat_amd.so libbeat_amd_F.so}; do
    BEATAMD_LIB=$R/beat_amd/$lib timeout 600 python tools/exp_variants.py $O/$lib.$rep.jsonl ${VARIANTS:-tools/variants_ab.json} > $O/$lib.$rep.log 2>&1
  done
done
