cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sum() { f=$(find gpurun_out/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" > gpurun_out/$1.txt 2>&1; rm -rf gpurun_out/$1; }
timeout 300 python -m pytest tests/test_fir_gpu.py tests/test_autograd_gpu.py -m gpu -q 2>&1 | tail -n 2
for r in 1 2; do
  for v in new old; do
    tag=r05_fir_ends_${v}_$r
    if [ $v = old ]; then export NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_firold.so; else unset NNMNKWII_AMD_SO; fi
    rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o run -- python tools/dbg/fir_run.py 64 500 60 > gpurun_out/$tag.log 2>&1
    sum $tag; echo "$v $r"; grep fir_kernel gpurun_out/$tag.txt | cut -c1-160
  done
done
unset NNMNKWII_AMD_SO
timeout 100 python tools/dbg/fir_soak.py 30 5 2>&1 | tail -n 1
