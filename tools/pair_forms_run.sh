# S2 decoded ITERS times with every variant library of tools/pair_forms.sh; prints the decodes that differ from the first one
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/probe; O=gpurun_out/probe/forms_${TAG:-a}.txt; : > $O
for f in ${FORMS:-1 2 3 4 5 6 0}; do
  if [ $f = 0 ]; then unset VIDC_LIBRARY; else export VIDC_LIBRARY=$GRAFT_REPO_ROOT/tools/_bin/libvidc_form$f.so; fi
  echo "== form $f" >> $O
  GPU_MAX_HW_QUEUES=8 NQS=${NQS:-4,8,5,6} ITERS=${ITERS:-200} timeout 900 python tools/repro_s2b.py "X=1" 2>&1 | grep -v "amdgpu.ids" | grep "differs\|it $((${ITERS:-200}-1)) \|Error\|error\|fault" | cut -c1-330 >> $O
done
cat $O
