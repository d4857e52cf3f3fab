export TMPDIR=/tmp; O=gpurun_out/r05bis; mkdir -p $O
run() { name=$1; shift; python -X faulthandler -m pytest "$@" -m gpu -q -x > $O/$name.txt 2>&1; echo "$name rc=$?" >> $O/rc.txt; tail -4 $O/$name.txt | head -3 >> $O/rc.txt; }
: > $O/rc.txt
run sharded_all tests/test_gpu_sharded.py
run sharded_virtual tests/test_gpu_sharded.py -k "virtual_shards or more_shards"
run sharded_rccl tests/test_gpu_sharded.py -k "rccl_all_gather"
run sharded_torch tests/test_gpu_sharded.py -k "torch_rccl"
run sharded_entry tests/test_gpu_sharded.py -k "every_entry_point and 1-1"
run two_tier tests/test_gpu_two_tier.py
cat $O/rc.txt
