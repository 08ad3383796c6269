CB_DEBUG=1 python bench.py --steps 30 --warmup 0 --no-cpu-baseline --no-secondary 2> gpurun_out/eig_dbg.err > gpurun_out/eig_dbg.out
grep -c "first-order" gpurun_out/eig_dbg.err
