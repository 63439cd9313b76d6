set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
python bench.py --mode rollout --steps 20 --warmup 5 --no-cpu-baseline > $O/base_rollout.json 2> $O/base_rollout.err
timeout -k 10 600 python -m pytest tests/test_parallel_gpu.py -x -q > $O/test_parallel.log 2>&1; echo "parallel rc=$?" >> $O/test_parallel.log
SEA_DUMP_MAPS=$O/maps_persist.txt rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kv -o run -- python3 bench.py --mode kv --steps 3 --warmup 1 > $O/prof_kv.json 2> $O/prof_kv.err; echo "kv persist rc=$?" > $O/prof_kv.rc
SEA_TUNE=kv_persist=0 SEA_DUMP_MAPS=$O/maps_nopersist.txt rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kv0 -o run -- python3 bench.py --mode kv --steps 3 --warmup 1 > $O/prof_kv0.json 2> $O/prof_kv0.err; echo "kv nopersist rc=$?" > $O/prof_kv0.rc
rm -rf $O/prof_kv/*trace* $O/prof_kv0/*trace*
tail -3 $O/test_parallel.log; cat $O/prof_kv.rc $O/prof_kv0.rc; tail -c 600 $O/base_rollout.json
