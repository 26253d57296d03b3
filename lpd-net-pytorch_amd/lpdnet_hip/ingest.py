"""Submap ingest: the reference reads every cloud with np.fromfile as 4096 x 3 float64, stacks a
batch, converts with `.float()` and moves it with `.to(device)` on the step's own stream (loading_pointclouds.py:26-47,
evaluate.py:111-117).  Here the raw float64 bytes are read into pinned host buffers, copied on a side HIP stream while the
previous batch is being embedded, and narrowed to float32 on the GPU (lpd_f64_to_f32): the forward never waits for the
host, and the host does no per-element work.

  load_pc_file / load_pc_files   loading_pointclouds.py:26-47, same return values (numpy float64)
  SubmapStream                   iterator over device batches [B,1,N,3] float32, double-buffered
  get_latent_vectors_from_files  evaluate.get_latent_vectors on a list of file names
"""
import os

import numpy as np
import torch

from . import ops

NUM_POINTS = 4096


def load_pc_file(filename, dataset_folder="", num_points=NUM_POINTS):
    """-> [num_points, 3] float64, or an empty array when the file does not hold num_points * 3 doubles (the reference logs
    "Error in pointcloud shape" and returns np.array([]))."""
    pc = np.fromfile(os.path.join(dataset_folder, filename), dtype=np.float64)
    if pc.shape[0] != num_points * 3:
        return np.array([])
    return np.reshape(pc, (pc.shape[0] // 3, 3))


def load_pc_files(filenames, dataset_folder="", num_points=NUM_POINTS):
    """-> [n_ok, num_points, 3] float64; files of the wrong size are skipped (loading_pointclouds.py:38-47)."""
    pcs = []
    for filename in filenames:
        pc = load_pc_file(filename, dataset_folder, num_points)
        if pc.shape[0] != num_points:
            continue
        pcs.append(pc)
    return np.array(pcs)


class SubmapStream:
    """for batch in SubmapStream(files, 32, folder, device): batch is a float32 CUDA tensor [b,1,N,3] on the CURRENT stream
    (b <= batch_size: ragged tail, wrong-size files skipped like load_pc_files).  Two pinned buffers and a copy stream:
    batch i+1 is read from disk and copied while batch i is consumed."""

    def __init__(self, filenames, batch_size, dataset_folder="", device=None, num_points=NUM_POINTS):
        self.files = list(filenames)
        self.bs, self.folder, self.N = int(batch_size), dataset_folder, int(num_points)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.pinned = [torch.empty((self.bs, self.N, 3), dtype=torch.float64).pin_memory() for _ in range(2)]
        self.staged = [torch.empty((self.bs, self.N, 3), dtype=torch.float64, device=self.device) for _ in range(2)]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]     # H2D copy of slot s finished
        self.free = [torch.cuda.Event(), torch.cuda.Event()]      # consumer finished reading slot s

    def _stage(self, slot, start):
        """read up to batch_size valid clouds starting at file index `start` into slot; -> (count, next file index)"""
        self.ready[slot].synchronize()        # the previous H2D copy out of this pinned buffer has left the host
        host = self.pinned[slot].numpy()
        n, i = 0, start
        while n < self.bs and i < len(self.files):
            pc = np.fromfile(os.path.join(self.folder, self.files[i]), dtype=np.float64)
            i += 1
            if pc.shape[0] != self.N * 3:
                continue
            host[n] = pc.reshape(self.N, 3)
            n += 1
        if n:
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(self.free[slot])          # the consumer of this device slot is done with it
                self.staged[slot][:n].copy_(self.pinned[slot][:n], non_blocking=True)
                self.ready[slot].record(self.copy_stream)
        return n, i

    def __iter__(self):
        main = torch.cuda.current_stream(self.device)
        for ev in self.free:
            ev.record(main)
        slot, pos = 0, 0
        n, pos = self._stage(slot, pos)
        while n:
            other = slot ^ 1
            n_next, pos = self._stage(other, pos) if pos < len(self.files) else (0, pos)   # disk + PCIe under the consumer
            main.wait_event(self.ready[slot])
            batch = ops.f64_to_f32(self.staged[slot][:n]).view(n, 1, self.N, 3)
            self.free[slot].record(main)
            yield batch
            slot, n = other, n_next


def get_latent_vectors_from_files(model, filenames, batch_size, dataset_folder="", output_dim=256, num_points=NUM_POINTS):
    """evaluate.py:96-159 on file names: eval mode, no_grad, batches of `batch_size` clouds streamed from disk, the previous
    train/eval mode restored afterwards (like harness.get_latent_vectors); -> numpy [n_ok, output_dim]."""
    was_training = model.training
    model.eval()
    outs = []
    dev = next(model.parameters()).device
    try:
        with torch.no_grad():
            for batch in SubmapStream(filenames, batch_size, dataset_folder, dev, num_points):
                outs.append(model(batch).detach().cpu().numpy().reshape(batch.shape[0], -1))
    finally:
        model.train(was_training)
    return np.concatenate(outs, 0) if outs else np.zeros((0, output_dim), np.float32)
