"""Does capturing the whole quadruplet train step (forward, lazy quadruplet loss, backward, Adam) in a HIP graph pay at the
reference's small batches?   python tools/train_graph.py [storage] [bq P Ng ...]     (default: f32, tuples (2,1,2) and (2,2,18))

The step has no host synchronisation (kNN graphs, CSR build, statistics and the loss all stay on the device), the statistics
workspace is capture-legal (ops._stat_ws), and torch's Adam runs captured with capturable=True.  Prints eager and replay ms per step
and checks that a replayed step produces the loss of the eager step on the same batch and weights."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from util.PointNetVlad import PointNetVlad
import loss.pointnetvlad_loss as L
from lpdnet_hip import autograd

dev = torch.device("cuda:0")
storage = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("f32", "bf16") else "f32"
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
shapes = [tuple(nums[i:i + 3]) for i in range(0, len(nums) - 2, 3)] or [(2, 1, 2), (2, 2, 18)]
autograd.set_train_storage(storage)


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for bq, P, Ng in shapes:
    B = bq * (1 + P + Ng + 1)
    torch.manual_seed(0)
    model = PointNetVlad(num_points=4096, featnet="lpdnet").to(dev).train()
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(7)
    batches = [(torch.rand((B, 1, 4096, 3), generator=gen) * 2 - 1).to(dev) for _ in range(8)]
    xs = torch.empty_like(batches[0])

    def make(capturable):
        model.load_state_dict(state0)
        opt = torch.optim.Adam(model.parameters(), lr=1e-5, fused=True, capturable=capturable)

        def step():
            opt.zero_grad(set_to_none=False)
            out = model(xs).view(bq, -1, 256)
            q, p, n, o = torch.split(out, [1, P, Ng, 1], dim=1)
            loss = L.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
            loss.backward()
            opt.step()
            return loss
        return opt, step

    # eager
    opt, step = make(False)
    it = [0]

    def eager():
        xs.copy_(batches[it[0] % 8]); it[0] += 1
        return step()
    t_eager = timeit(eager)
    model.load_state_dict(state0)
    opt, step = make(False)
    ref_losses = []
    for i in range(3):
        xs.copy_(batches[i]); ref_losses.append(step().item())
    # graph
    opt, step = make(True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):
            xs.copy_(batches[i]); step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            loss_g = step()
    except Exception as exc:      # noqa: BLE001
        print(f"bq={bq} P={P} Ng={Ng} ({B} clouds, {storage}): eager {t_eager:.3f} ms; capture failed: {type(exc).__name__}: {str(exc)[:300]}")
        continue
    # replays from the initial weights: the same three batches must give the eager losses
    model.load_state_dict(state0)
    for st in opt.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    got = []
    for i in range(3):
        xs.copy_(batches[i]); g.replay(); got.append(loss_g.item())

    def replay():
        xs.copy_(batches[it[0] % 8]); it[0] += 1
        g.replay()
    t_graph = timeit(replay)
    print(f"bq={bq} P={P} Ng={Ng} ({B} clouds, {storage}): eager {t_eager:.3f} ms/step, graph replay {t_graph:.3f} ms/step; "
          f"losses eager {['%.5f' % v for v in ref_losses]} graph {['%.5f' % v for v in got]}")
