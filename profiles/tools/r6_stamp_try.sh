# Round 6: build the library HERE with -DCB_EIGH_STAMPS + the given flags, run the headline window on the GPU box with CB_DEBUG=1 and
# print what each launch of a planned solve costs.   bash profiles/tools/r6_stamp_try.sh LABEL [extra hipcc flags]
L=$1; shift
export CB_EXTRA_HIPCC_FLAGS="-DCB_EIGH_STAMPS $*"
python -c "from cherryml_amd import _build; _build.build()" 2>&1 | grep -i "warning\|error\|built" | tail -3
/usr/local/graft/bin/gpurun --timeout 300 -- "mkdir -p gpurun_out/r6e; export CB_EXTRA_HIPCC_FLAGS='$CB_EXTRA_HIPCC_FLAGS'; CB_DEBUG=1 timeout 200 python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2> gpurun_out/r6e/stamps_$L.txt > /dev/null < /dev/null; python3 profiles/tools/r6_eigh_stamps.py gpurun_out/r6e/stamps_$L.txt 20 | python3 -c \"
import json,sys; d=json.load(sys.stdin); print(d['solve_us_mean'], d['solve_us_min_max'], d['begin_plus_warm_start_us']); [print(k, v) for k,v in d['launches_that_ran'].items()]\"" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)"
