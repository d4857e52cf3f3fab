import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, frankensearch_amd as fa
dev = torch.device("cuda", 0)
rows = 10_000_000
slab = bench.gen_corpus(0, rows, 384, dev)
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, 384, device=0, keepalive=slab)
if len(sys.argv) > 1: idx.set_batched_filter(int(sys.argv[1]))
tt = bench.two_tier_section(idx, rows, 10, dev, 0)
for k, v in tt.items():
    if k.startswith("concurrent"): print(k, json.dumps(v)[:700])
