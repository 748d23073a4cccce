cd $GRAFT_REPO_ROOT
for v in "" "$@"; do  # usage: tools/ab_chain.sh -DVARIANT_A -DVARIANT_B (run through gpurun)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -I vector_db_id_compression_amd/csrc tools/prof_u20.hip -o /tmp/pp 2>&1 | grep error
  echo "variant [$v]"
  cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/pv -o v -- /tmp/pp > /dev/null 2>&1
  python3 - <<'PY'
import sqlite3,glob
db=sqlite3.connect(glob.glob('/tmp/pv/*results.db')[0])
for r in db.execute("select name, average from top_kernels where name like '%encode_u%'"): print('   ', r[0][:40], round(r[1]/50000*1000,1), 'ns/step')
PY
  rm -rf /tmp/pv; cd $GRAFT_REPO_ROOT
done
