import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
import test_train_gpu as T
DEV = "cuda"
def run(graphed):
    ts, videos, tokens, lengths, noise = T._build_step()
    args = (videos.to(DEV), tokens.to(DEV), lengths.to(DEV))
    f = ts.step_graphed if graphed else ts.step
    losses = [f(*args, init_noise=noise.to(DEV)) for _ in range(3)]
    l = [float(x["loss"]) for x in losses]
    return l, {n: v.data.detach().cpu().clone() for n, v in ts.model.names.items()}
def cmp(a, b, tag):
    worst = max(((a[1][n] - b[1][n]).abs().max().item(), n) for n in a[1])
    cnt = sum(int(((a[1][n] - b[1][n]).abs() > 1e-5).sum()) for n in a[1])
    print(tag, "losses", a[0], b[0], "worst", worst, "count>1e-5", cnt, flush=True)
runs = [run(False) for _ in range(int(os.environ.get("PROBE_RUNS", "4")))]
for i in range(1, len(runs)):
    cmp(runs[0], runs[i], f"eager0-eager{i}")
if os.environ.get("PROBE_GRAPH", "0") == "1":
    g1, g2 = run(True), run(True)
    cmp(g1, g2, "graph-graph")
    cmp(runs[0], g1, "eager-graph")
