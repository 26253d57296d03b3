"""MI355X-native LPD-Net feature networks behind the reference's module API.

Mirror of the hot-path part of util/lpdnet_model.py of qiaozhijian/LPD-Net-Pytorch: the same public
names, constructor signatures, tensor shapes and state_dict keys (so checkpoints and the reference's
train/eval scripts drop in), but every forward pass runs on the gfx950 kernels of liblpd_hip.so
(lpdnet_hip/engine.py for inference, lpdnet_hip/autograd.py for training).  The nn.Conv*/nn.BatchNorm*
children below are parameter containers only -- their own forward() is never called.

Differences from the reference, on purpose:
  * no pynvml / MemTracker import side effects, no hard-coded torch.device('cuda')
    (reference lpdnet_model.py:10-14,123,307,338);
  * `k` is a constructor keyword (hard-coded 20 at reference lpdnet_model.py:31,156);
  * `use_mFea=True` takes [B,1,N,8] inputs (xyz + 5 features) like the reference; the points are then not Z-ordered internally;
  * inputs must live on the GPU: there is no CPU fallback.
The dead registration model (`LPD`, reference lpdnet_model.py:366-582) is out of scope.
"""
import torch
import torch.nn as nn

from lpdnet_hip import engine, ops

cat_or_stack = True  # reference lpdnet_model.py:17 (the stack variant is dead code there)


def knn(x, k):
    """Reference lpdnet_model.py:317-326: x [B,C,N] -> int64 indices [B,N,k] of the k nearest points
    (self included), nearest first.  Bit-exact with the reference CPU path on tie-free rows."""
    return ops.knn(x.float(), k).long()


def _gather_neighbours(x, k, idx):
    B, N = x.size(0), x.size(2)
    x = x.reshape(B, -1, N)
    if idx is None:
        idx = knn(x, k)
    pts = x.transpose(2, 1).contiguous()                        # [B,N,C]
    flat = (idx + torch.arange(B, device=x.device).view(-1, 1, 1) * N).reshape(-1)
    nbr = pts.reshape(B * N, -1)[flat].view(B, N, idx.shape[-1], -1)
    return pts, nbr


def get_graph_feature(x, k=20, idx=None):
    """Reference lpdnet_model.py:331-363: [B,C,N(,1)] -> edge tensor [B,2C,N,k] = cat(neighbour, centre).

    Kept for API parity (it materialises the edge tensor; the model itself never does -- see
    csrc/lpd_edge.hip).  The kNN runs on the HIP kernel; the gather is plain device indexing."""
    pts, nbr = _gather_neighbours(x, k, idx)
    ctr = pts.unsqueeze(2).expand_as(nbr)
    return torch.cat((nbr, ctr), dim=3).permute(0, 3, 1, 2)


def get_graph_feature_Origin(x, k=20, idx=None, cat=True):
    """Reference lpdnet_model.py:116-145: cat(centre, neighbour - centre) [B,2C,N,k], or the
    neighbours alone [B,C,N,k] when cat=False."""
    pts, nbr = _gather_neighbours(x, k, idx)
    if cat:
        ctr = pts.unsqueeze(2).expand_as(nbr)
        return torch.cat((ctr, nbr - ctr), dim=3).permute(0, 3, 1, 2)
    return nbr.permute(0, 3, 1, 2)


def _channel_major(module, feat, B, N):
    """point-major rows -> [B,E,N,1]; in train mode through an autograd node so that the public trunk output carries gradients"""
    if module.training and feat.requires_grad:
        from lpdnet_hip import autograd
        return autograd.to_channel_major_train(feat, B, N)
    return engine.to_channel_major(feat, B, N)


def _make_act(use_relu, slope):
    return nn.ReLU(inplace=True) if use_relu else nn.LeakyReLU(negative_slope=slope, inplace=True)


class TranformNet(nn.Module):
    """T-Net (reference lpdnet_model.py:273-313): [B,k,N] -> [B,k,k] alignment matrix."""

    def __init__(self, k=3, negative_slope=1e-2, use_relu=True):
        super().__init__()
        self.conv1 = nn.Conv1d(k, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k * k)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)
        self.k = k

    def forward(self, x):
        if x.dim() != 3 or x.shape[1] != self.k:
            raise ValueError(f"TranformNet(k={self.k}): expected [B,{self.k},N], got {tuple(x.shape)}")
        B, N = x.shape[0], x.shape[2]
        if self.training:           # standalone train-mode forward (batch statistics, autograd): reference lpdnet_model.py:295-313
            from lpdnet_hip import autograd
            rows = x.float().transpose(1, 2).reshape(B * N, self.k)      # [B,k,N] -> point-major rows (torch op: carries the gradient)
            return autograd.tnet_train(self, rows.contiguous(), B, N, use_bn=True)
        rows = ops.transpose(x.float().contiguous()).view(B * N, self.k)
        return engine.transform_net_eval(self, rows, B, N)


class LPDNet(nn.Module):
    """Reference lpdnet_model.py:147-268.  forward: [B,1,N,3] -> [B,emb_dims,N,1]."""

    def __init__(self, emb_dims=512, use_mFea=False, t3d=True, tfea=False, use_relu=False, k=20):
        super().__init__()
        self.negative_slope = 1e-2
        self.use_relu = use_relu
        self.act_f = _make_act(use_relu, self.negative_slope)
        self.use_mFea = use_mFea
        self.k = k
        self.t3d = t3d
        self.tfea = tfea
        self.emb_dims = emb_dims
        if self.t3d:
            self.t_net3d = TranformNet(3)
        if self.tfea:
            self.t_net_fea = TranformNet(64)
        self.useBN = True
        self.convDG1 = nn.Sequential(nn.Conv2d(128, 128, kernel_size=1, bias=False), nn.BatchNorm2d(128), self.act_f)
        self.convDG2 = nn.Sequential(nn.Conv2d(128, 128, kernel_size=1, bias=False), nn.BatchNorm2d(128), self.act_f)
        self.convSN1 = nn.Sequential(nn.Conv2d(256, 256, kernel_size=1, bias=False), nn.BatchNorm2d(256), self.act_f)
        # use_mFea: [B,1,N,8] inputs = xyz + 5 handcrafted features (reference lpdnet_model.py:183-186,215-224)
        self.conv1_lpd = nn.Conv1d(8 if use_mFea else 3, 64, kernel_size=1, bias=False)
        self.conv2_lpd = nn.Conv1d(64, 64, kernel_size=1, bias=False)
        self.conv3_lpd = nn.Conv1d(512, self.emb_dims, kernel_size=1, bias=False)
        self.bn1_lpd = nn.BatchNorm1d(64)
        self.bn2_lpd = nn.BatchNorm1d(64)
        self.bn3_lpd = nn.BatchNorm1d(self.emb_dims)

    def _features(self, x, reorder=True):
        """Point-major features ([B*N, E], B, N): what NetVLADLoupe consumes without a layout change.  reorder: the points of
        each cloud may be Z-ordered internally (rows of the result then follow that order, not the caller's) -- legal for
        PointNetVlad, whose descriptor is invariant to point order."""
        if self.training:
            from lpdnet_hip import autograd
            return autograd.lpdnet_features_train(self, x, reorder)
        return engine.lpdnet_features_eval(self, x, reorder)

    def forward(self, x):
        feat, B, N = self._features(x, reorder=False)      # column n of the result belongs to input point n, like the reference
        return _channel_major(self, feat, B, N)


class LPDNetOrign(nn.Module):
    """Reference lpdnet_model.py:18-114 (the argparse-default `featnet='lpdnetorigin'`)."""

    def __init__(self, emb_dims=512, use_mFea=False, t3d=True, tfea=False, use_relu=False, k=20):
        super().__init__()
        self.negative_slope = 1e-2
        self.use_relu = use_relu
        self.act_f = _make_act(use_relu, self.negative_slope)
        self.use_mFea = use_mFea
        self.k = k
        self.t3d = t3d
        self.tfea = tfea
        self.emb_dims = emb_dims
        if self.t3d:
            self.t_net3d = TranformNet(3)
        if self.tfea:
            self.t_net_fea = TranformNet(64)
        self.useBN = True

        def c2(i, o):
            return nn.Sequential(nn.Conv2d(i, o, kernel_size=1, bias=False), nn.BatchNorm2d(o), self.act_f)

        def c1(i, o):
            return nn.Sequential(nn.Conv1d(i, o, kernel_size=1, bias=False), nn.BatchNorm1d(o), self.act_f)
        self.convDG1 = c2(128, 64)
        self.convDG2 = c2(64, 64)
        self.convSN1 = c2(64, 64)
        self.convSN2 = c2(64, 64)
        self.conv1_lpd = c1(8 if use_mFea else 3, 64)
        self.conv2_lpd = c1(64, 64)
        self.conv3_lpd = c1(64, 64)
        self.conv4_lpd = c1(64, 128)
        self.conv5_lpd = c1(128, self.emb_dims)

    def _features(self, x, reorder=True):
        if self.training:
            from lpdnet_hip import autograd
            return autograd.lpdnet_origin_features_train(self, x, reorder)
        return engine.lpdnet_origin_features_eval(self, x, reorder)

    def forward(self, x):
        feat, B, N = self._features(x, reorder=False)      # per-point output in the caller's point order
        return _channel_major(self, feat, B, N)
