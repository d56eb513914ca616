cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bj
timeout -k 10 1000 python tests/fuzz_parity.py 10000 20261004 > gpurun_out/r02bj/fuzz.txt 2>&1; rc=$?
F=gpurun_out/r02bj/fuzz.txt
{ echo "# tests/fuzz_parity.py 10000 20261004 on the round's final build (libtyrant_hip_diag.so): summary"
  echo "ok cases: $(grep -c ' -> ok' $F)"
  echo "failures: $(grep -c 'FAIL' $F)"
  echo "sharded (nranks > 1): $(grep ' -> ' $F | grep -vc 'rank 0/1 ')"
  echo "with emissive triangles: $(grep ' -> ' $F | grep -c '+lights')"
  echo "with colour palettes: $(grep ' -> ' $F | grep -c '+colors')"
  echo "merged trace launches: $(grep ' -> ' $F | grep -c "'merge_trace': 1")"
  echo "... of those one iteration ahead of the counts: $(grep ' -> ' $F | grep "'merge_trace': 1" | grep -c "'run_ahead': 1")"
  echo "... of those with the wide drain: $(grep ' -> ' $F | grep "'merge_trace': 1" | grep -c "'wide_drain': 1")"
  echo "variant 5: $(grep ' -> ' $F | grep -c "'traversal_variant': 5")"
  echo "# first 8 cases:"
  grep ' -> ' $F | head -8
} > gpurun_out/r02bj/fuzz_summary.txt
head -10 gpurun_out/r02bj/fuzz_summary.txt
rm -f $F
exit $rc
