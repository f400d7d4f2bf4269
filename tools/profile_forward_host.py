import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lanegcn_amd
from lanegcn_amd import data as gen, lanegcn as M
torch.manual_seed(0)
net = M.Net(M.config).cuda().eval()
batch = gen.collate_fn(gen.synth_batch("S2", seed=5))
with torch.no_grad():
    for _ in range(5):
        out = net(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = net(batch)
    torch.cuda.synchronize()
    print("net(data) no_grad: %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        out = net(batch)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
