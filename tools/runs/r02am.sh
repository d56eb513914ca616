cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02am
timeout -k 10 1000 python tests/fuzz_parity.py 4000 7 > gpurun_out/r02am/fuzz_4000.txt 2>&1; echo "rc $?"
F=gpurun_out/r02am/fuzz_4000.txt
{ echo "# tests/fuzz_parity.py 4000 7 on the round-2 build (libtyrant_hip_diag.so): summary"
  echo "ok cases: $(grep -c ' -> ok' $F)"
  echo "failures: $(grep -c 'FAIL' $F)"
  echo "sharded (nranks > 1): $(grep ' -> ' $F | grep -vc 'rank 0/1 ')"
  echo "with emissive triangles: $(grep ' -> ' $F | grep -c '+lights')"
  echo "with colour palettes: $(grep ' -> ' $F | grep -c '+colors')"
  echo "merged trace launches: $(grep ' -> ' $F | grep -c "'merge_trace': 1")"
  echo "variant 5: $(grep ' -> ' $F | grep -c "'traversal_variant': 5")"
  echo "# first 12 cases:"
  grep ' -> ' $F | head -12
} > gpurun_out/r02am/fuzz_4000_summary.txt
cat gpurun_out/r02am/fuzz_4000_summary.txt | head -9
rm -f $F.gz; gzip -k $F
