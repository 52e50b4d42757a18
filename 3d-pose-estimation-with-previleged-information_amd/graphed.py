"""Whole-step HIP-graph capture (opt-in, single process): one `hipGraphLaunch` per training step instead of ~600 kernel launches.

Measured on MI355X / ROCm 7.2 (tools/graph_try.py, ResNet-50 bs 64): the fp32 step replays in 43.1 ms against 43.6 ms eager (+1 %),
the fp16 step in 17.9 ms against 16.0 ms eager (SLOWER: the runtime launches the ~600 kernel nodes of the graph one by one with a
dependency barrier each, which costs more than the stream launches it replaces).  So this stays an opt-in experiment; it documents
that the step IS capturable and keeps the property tested.  What makes the step capturable: no host read-back anywhere (loss, overflow test and Adam step counter stay on the device), caller-owned workspaces,
parameters / gradients / moments at fixed addresses (FlatAdam), and the second (wgrad) stream forks from and joins the capturing
stream inside the step.  Values that are baked into the captured launches -- the learning rate and, for the fp32 optimizer, the
host-side step count -- are handled by re-capturing when the learning rate changes and by using the device-side step counter
(`FlatAdam.clip_and_step_dev`) for both precisions.

Not used under torch.distributed (the bucket hooks are Python callbacks that a replay does not run).
"""
import torch

from . import ops


class GraphedStep:
    """Wraps Trainer.train_step for fixed-shape batches: `step(color, depth, true_cam, true_val)` copies the batch into static
    buffers and replays the captured graph; returns the static loss tensor (device, valid until the next call)."""

    def __init__(self, trainer, warmup=3):
        warmup = max(int(warmup), 1)       # (at least one eager step: it builds the plans, the weight-image job table and sizes every workspace OUTSIDE the capture)
        if trainer.world != 1 or trainer.reducer.active:
            raise RuntimeError('GraphedStep: whole-step capture is single-process only')
        self.trainer = trainer
        self.warmup = warmup
        self.graph = None
        self.static = None
        self.loss = None
        self._lr = None
        self._shapes = None

    def _capture(self, batch):
        tr = self.trainer
        self.static = tuple(None if t is None else t.clone() for t in batch)
        run = self._run
        if self.graph is not None:
            # re-capture: the events inside the buffer sets were last recorded in the old graph; eager warm-up steps must not wait on them
            torch.cuda.synchronize()
            from . import ops_block
            ops_block.reset_side_events(tr.model)
            self.graph = None
        # the eager warm-up steps must not train: parameters, Adam moments, step counters and BatchNorm buffers are put back afterwards,
        # so the first call of step() amounts to exactly one optimisation step (the first replay), like every later call
        opt = tr.optimizer
        if opt.dev_state is None:
            opt.dev_state = torch.tensor([opt.step_count, 0], dtype=torch.int32, device=opt.flat_p.device)
            opt._dev_scratch = torch.zeros(4, dtype=torch.float32, device=opt.flat_p.device)
        saved = [t.clone() for t in (opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.dev_state)]
        saved_buffers = [b.clone() for b in tr.model.buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # eager warm-up on a side stream, as the PyTorch capture recipe asks
            for _ in range(self.warmup):
                run()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for dst, src in zip((opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.dev_state), saved):
                dst.copy_(src)
            for dst, src in zip(tr.model.buffers(), saved_buffers):
                dst.copy_(src)
        ops.weights_changed()          # the restore wrote flat_p under the parameter views; the capture below must contain the image rebuild of every weight
        # Build the batched weight-image job table for "every convolution stale" NOW: its first use uploads the table from the host, which a capture does not
        # allow (a single warm-up step only ever saw one-convolution tables, as each layer asked for its image for the first time).  Then mark the images stale
        # again so that the captured step contains the rebuild.
        from . import ops_block
        ops_block._rebuild_stale_images(opt.flat_p.device)
        ops.weights_changed()
        if tr.half_acc:
            from . import ops_half
            ops_half.refresh_weights(tr.model, opt.flat_p)
        torch.cuda.synchronize()
        # The buffer sets' "last weight-gradient reader" events were recorded by the warm-up steps (or inside a PREVIOUS capture, when this is a re-capture after
        # a learning-rate change): the device is idle now, so nothing has to wait for them, and a captured event must not be waited on from outside its graph.
        from . import ops_block
        ops_block.reset_side_events(tr.model)
        profile, ops.PROFILE = ops.PROFILE, None           # timing events cannot be recorded inside a capture
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = run()
        ops.PROFILE = profile
        self._lr = tr.optimizer.param_groups[0]['lr']
        self._shapes = tuple(None if t is None else tuple(t.shape) for t in batch)

    def _run(self):
        tr = self.trainer
        keep = tr.optimizer.clip_and_step
        # the host-counted fp32 optimizer call would bake its step number into the graph: route it to the device-side counter
        tr.optimizer.clip_and_step = lambda max_norm, grad_scale=1.0: tr.optimizer.clip_and_step_dev(max_norm, grad_scale, skip_nonfinite=False)
        try:
            return tr.train_step(*self.static)
        finally:
            tr.optimizer.clip_and_step = keep

    def step(self, color_image, depth_image, true_cam, true_val):
        batch = (color_image, depth_image, true_cam, true_val)
        shapes = tuple(None if t is None else tuple(t.shape) for t in batch)
        if self.graph is None or shapes != self._shapes or self.trainer.optimizer.param_groups[0]['lr'] != self._lr:
            self._capture(batch)                           # warm-up steps are rolled back; the capture pass only records
        else:
            for dst, src in zip(self.static, batch):
                if dst is not None:
                    dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # the replayed optimizer kernels changed the weights behind Python's back: whatever eager code runs next (an evaluation pass, the warm-up of a re-capture)
        # must rebuild the weight images (the graph itself rebuilds them at its head on every replay)
        ops.weights_changed()
        return self.loss
