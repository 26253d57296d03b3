"""T-Net standalone accuracy: MI355X train-mode forward vs the oracle in fp32 / fp64, layer by layer."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import lpd_oracle as orc, synth
from lpdnet_hip import autograd as ag, ops
from util.lpdnet_model import TranformNet

torch.manual_seed(0)
for kd in (3, 64):
    for B, N in ((6, 256), (6, 4096)):
        net = TranformNet(k=kd)
        sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_like({k: tuple(v.shape) for k, v in net.state_dict().items()}).items()}
        net.load_state_dict(sd)
        net = net.cuda().train()
        x = torch.randn(B, kd, N)
        if kd == 64:
            x = torch.nn.functional.leaky_relu(x + 0.3)
        rows = x.transpose(1, 2).contiguous().view(B * N, kd).cuda()
        with torch.no_grad():
            t, S = ag._TNet.fwd(net, rows, B, N, True)
        ref = {}
        for dt in (torch.float32, torch.float64):
            sdd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
            ref[dt] = orc.transform_net(sdd, "", x.to(dt), True, {}).double()
        def e(a, b):
            return ((a - b).abs().max() / b.abs().max()).item()
        print(kd, B, N, "gpu-64 %.2e gpu-32 %.2e 32-64 %.2e" % (e(t.cpu().double(), ref[torch.float64]), e(t.cpu().double(), ref[torch.float32]),
              e(ref[torch.float32], ref[torch.float64])), "|t|max %.2f" % ref[torch.float64].abs().max().item(), flush=True)
        # layerwise vs fp64
        import torch.nn.functional as F
        sdd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        h = x.double()
        for i, name in enumerate(("1", "2", "3")):
            raw = orc._conv1x1(sdd, "conv" + name, h)
            h = torch.relu(orc._bn(sdd, "bn" + name, raw, True, {}))
            got = S["r" + name].cpu().double().view(B, N, -1).transpose(1, 2)
            print("   conv%s raw err %.2e" % (name, e(got, raw)), end="")
        g = h.max(dim=2)[0]
        print("   pooled err %.2e" % e(S["g"].cpu().double(), g), end="")
        raw4 = orc._linear(sdd, "fc1", g)
        print("   fc1 raw %.2e" % e(S["r4"].cpu().double(), raw4), end="")
        a4 = torch.relu(orc._bn(sdd, "bn4", raw4, True, {}))
        print("   a4 %.2e" % e(S["a4"].cpu().double(), a4), end="")
        raw5 = orc._linear(sdd, "fc2", a4)
        a5 = torch.relu(orc._bn(sdd, "bn5", raw5, True, {}))
        print("   a5 %.2e" % e(S["a5"].cpu().double(), a5))
