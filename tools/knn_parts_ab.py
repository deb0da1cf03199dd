import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import bench_ops
for b, p in ((64, 2048), (128, 1024), (16, 1024), (8, 2048)):
    print("batch", b, "points", p, "parts", os.environ.get("DGNN_KNN_PARTS"), flush=True)
    bench_ops.knn(batch=b, pts=p)
