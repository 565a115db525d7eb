"""UNet(nn.Module) with the reference's constructor, module tree and state_dict keys
(lib/UNet.py:104-246), executed by hand-written gfx950 kernels through libresdepth_hip.so.

The sub-modules (nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d ...) are kept ONLY as
parameter containers so that `state_dict()` / `load_state_dict()` / `parameters()` /
`.to(device)` / `.train()` / `.eval()` behave exactly like the reference's module and
published `.pth` checkpoints load unchanged.  `forward()` never calls them: it runs the whole
network as one autograd.Function over NHWC fp32 activations:

    encoder level i : conv3x3 -> BN stats -> BN-apply+act+maxpool(+argmax)      (lib/UNet.py:201-207)
    bottleneck      : conv3x3 -> BN stats -> BN-apply+act                       (lib/UNet.py:210)
    decoder level i : convT2x2 (+bias, +skip ADD fused) -> conv3x3 -> BN -> act (lib/UNet.py:213-224)
    head            : conv3x3 C->1 (+bias) + x[:,0:1]                           (lib/UNet.py:227-244)

There is no CPU fallback: inputs must live on a HIP device.
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch
import torch.nn as nn

from . import _lib, ops

_SLOPES = {"relu": 0.0, "lrelu": 0.01}


def _check_valid_activation(choice):
    # same message / exception type as lib/UNet.py:12-14
    if choice not in ["relu", "lrelu", "prelu"]:
        raise ValueError(f"'{choice}' is not a valid activation function. Choose among ['relu', 'lrelu', 'prelu'].\n")


def _make_activation(choice):
    # the reference builds all three and returns one (lib/UNet.py:27-33); none of them draws random numbers
    return {"relu": nn.ReLU(inplace=True), "lrelu": nn.LeakyReLU(inplace=True), "prelu": nn.PReLU()}[choice]


def _conv3x3(cin, cout, bias):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=bias)


def _upconv(c, mode):
    # RNG-order contract: the reference's upconv() constructs the bilinear branch's
    # conv1x1 (weight + bias draws) BEFORE the ConvTranspose2d and discards it (lib/UNet.py:17-24).
    discarded = nn.Conv2d(c, c, kernel_size=1, stride=1)
    up = nn.ConvTranspose2d(c, c, kernel_size=2, stride=2)
    if mode == "bilinear":
        return nn.Sequential(nn.Upsample(mode="bilinear", scale_factor=2), discarded)
    return up


def _block(cin, cout, act, do_bn):
    if do_bn:
        return nn.Sequential(_conv3x3(cin, cout, False), nn.BatchNorm2d(cout), _make_activation(act))
    return nn.Sequential(_conv3x3(cin, cout, True), _make_activation(act))


class _Packed:
    """Packed weight operands per layer + the event that marks each one ready (see UNet._packed)."""

    def __init__(self):
        self.items, self.events = {}, {}
        self._waited = set()             # (event, stream) pairs already ordered: the fused pack marks EVERY layer with one event

    def get(self, key):
        ev = self.events.pop(key, None)
        if ev is not None:
            cur = torch.cuda.current_stream()
            mark = (id(ev), cur.cuda_stream)
            if mark not in self._waited:
                # one wait per (event, stream): a wait on an event that has long fired still costs the queue a barrier packet --
                # ~6 us in front of every convolution of the forward pass (kernel trace, profiles/r06_notes.md section 13)
                self._waited.add(mark)
                self._keep = getattr(self, "_keep", []) + [ev]       # ids stay unique while the marks live
                _lib.ev_wait(cur, ev)
        return self.items[key]


class SkipConnection(nn.Module):
    """Parameter-free marker kept for module-tree parity (lib/UNet.py:96-101); the ADD itself is fused
    into the transposed-convolution epilogue."""

    def forward(self, x_skip, x_up):
        return x_skip + x_up


class UNet(nn.Module):
    def __init__(self, n_input_channels=1, start_kernel=64, max_filter_depth=512, depth=8,
                 act_fn_encoder="relu", act_fn_decoder="relu", act_fn_bottleneck="relu", up_mode="transpose",
                 do_BN=True, bias_conv_layer=False, outer_skip=True, outer_skip_BN=False):
        super().__init__()
        _check_valid_activation(act_fn_encoder)
        _check_valid_activation(act_fn_decoder)
        _check_valid_activation(act_fn_bottleneck)
        if up_mode not in ["transpose", "bilinear"]:
            raise ValueError(f"'{up_mode}' is not a valid mode for upsampling. Choose among ['transpose', 'bilinear'] "
                             "to specify 'up_mode'.\n")
        self.n_input_channels = n_input_channels
        self.start_kernel = start_kernel
        self.depth = depth
        self.act_fn_encoder = act_fn_encoder
        self.act_fn_decoder = act_fn_decoder
        self.act_fn_bottleneck = act_fn_bottleneck
        self.up_mode = up_mode
        self.max_filter_depth = max_filter_depth
        self.do_BN = do_BN
        self.bias_conv_layer = bias_conv_layer
        self.do_outer_skip = outer_skip
        self.do_outer_skip_BN = outer_skip_BN
        self.filter_depths = [min(start_kernel * (2 ** i), max_filter_depth) for i in range(depth)]
        fd = self.filter_depths

        # module tree: same names, same construction (= RNG draw) order as lib/UNet.py:157-194
        self.encoder = nn.ModuleList()
        cin = n_input_channels
        for c in fd:
            self.encoder.append(nn.Sequential(_block(cin, c, act_fn_encoder, do_BN), nn.MaxPool2d(kernel_size=2, stride=2)))
            cin = c
        self.bottleneck = _block(fd[-1], fd[-1], act_fn_bottleneck, do_BN)
        self.decoder = nn.ModuleList()
        self.filter_depths_up = list(reversed(fd))
        for ci, co in zip(self.filter_depths_up[:-1], self.filter_depths_up[1:]):
            self.decoder.append(nn.Sequential(_upconv(ci, up_mode), _block(ci, co, act_fn_decoder, do_BN)))
        self.decoder.append(_upconv(self.filter_depths_up[-1], up_mode))
        self.last_layer = _conv3x3(start_kernel, 1, bias_conv_layer)
        self.skipconnect = SkipConnection()
        if self.do_outer_skip:
            self.layer_outer_skip = nn.ModuleList()
            if self.do_outer_skip_BN:
                self.layer_outer_skip.append(nn.BatchNorm2d(1))
            self.layer_outer_skip.append(SkipConnection())

        # ---- engine state (not part of the state_dict) ----
        self._flat_param = None      # flat fp32 buffer holding every parameter (views are the nn.Parameters)
        self._flat_grad = None       # flat fp32 buffer the backward writes gradients into
        self._offsets = None
        self._pack_cache = None
        self._pack_key = None
        self.grad_sync = None        # optional resdepth_amd.dp.GradSync (data-parallel hooks)
        self.sync_bn = False
        self.two_stream_backward = True   # weight gradients on a second HIP stream, overlapping the dgrad/BN chain
        self.defer_wgrad_levels = int(os.environ.get("RD_DEFER_WGRAD", "1"))    # encoder levels 1..n launch their weight gradient behind their data gradient (see _engine_backward_impl)
        self.fold_eval_bn = True          # inference: eval-mode BN folded into the packed conv weights + epilogue activation
        self.fused_bn_bwd_stats = True    # BN-backward sums from the epilogues of the kernels producing the gradient operands
        self.composed_tail = True         # last up-convolution's gradients straight from the 1-channel dout (ops.tail_*)
        self.fused_first_wgrad = True     # level 0: BN / activation / pool backward evaluated inside the first conv's weight gradient
        self.fused_first_eval = True      # inference: level 0's BN + activation + max-pool inside the first convolution (no z0)
        self.fast_eval = False            # inference with ONE scale per tensor (as in training): results then depend on the batch composition in the last bits
        self.eval_per_image = True        # inference on the three-product (split2h) bodies with one scale per IMAGE: a tile's result depends on that tile alone (False: six-product bodies, r01-r05)
        # loss.backward(retain_graph=True) on the reference keeps the saved activations for a second backward.  A custom
        # autograd.Function cannot see that flag, and keeping 137 MB per tile alive until the loss tensor dies would surprise
        # callers that hold on to losses -- so the activations are released by the first backward unless this is set
        self.retain_activations = False
        self._bn_gen = 0                  # bumped by every training-mode forward (the kernels update running statistics in place)
        self._side_stream = None

    # ------------------------------------------------------------------------------------------
    def _first_generic(self) -> bool:
        """More than 6 input channels: the first convolution runs as an ordinary NHWC conv3x3 (MFMA-class kernels, packed
        operands) behind an NCHW -> NHWC transpose instead of the dedicated few-channel kernels."""
        return self.n_input_channels > 6

    def _needs_twin(self) -> bool:
        """The kernels move channels in groups of four.  Constructor arguments outside that domain (start_kernel or
        max_filter_depth not a multiple of 4, more than 6 input channels not a multiple of 4) run on a zero-padded twin
        (`_twin`), which computes the same function."""
        if any(c % 4 for c in self.filter_depths):
            return True
        return self.n_input_channels > 6 and self.n_input_channels % 4 != 0

    def _twin(self):
        """The channel-padded engine model of a UNet whose widths the kernels do not take directly: same architecture with
        start_kernel / max_filter_depth / (n_input_channels > 6) rounded up to multiples of 4 (every layer then has at
        least as many channels as this model's).  Parameters live in the leading channels; the padding channels carry
        zero weights, BN (gamma, beta) = (1, 0) and zero biases, so their activations are exactly 0 in training and eval
        mode and they neither influence a real channel nor receive a gradient through one."""
        tw = self.__dict__.get("_twin_model")
        dev = next(self.parameters()).device
        if tw is None or next(tw.parameters()).device != dev:
            up4 = lambda v: (v + 3) // 4 * 4
            cin = up4(self.n_input_channels) if self.n_input_channels > 6 else self.n_input_channels
            with torch.random.fork_rng(devices=[]):        # the twin's default init must not advance the caller's RNG stream
                tw = UNet(cin, up4(self.start_kernel), up4(self.max_filter_depth), self.depth, self.act_fn_encoder,
                          self.act_fn_decoder, self.act_fn_bottleneck, self.up_mode, self.do_BN, self.bias_conv_layer,
                          self.do_outer_skip, self.do_outer_skip_BN).to(dev)
            with torch.no_grad():
                for name, t in list(tw.named_parameters()) + list(tw.named_buffers()):
                    if name.endswith("num_batches_tracked"):
                        continue
                    is_bn_scale = t.dim() == 1 and (name.endswith(".1.weight") or name.endswith("running_var")
                                                    or name == "layer_outer_skip.0.weight")
                    t.fill_(1.0 if is_bn_scale else 0.0)
            for q in tw.parameters():
                q.requires_grad_(False)              # gradients come back through _TwinFunction, never through autograd
            self.__dict__["_twin_model"] = tw         # not a registered sub-module: invisible to state_dict / parameters
        return tw

    @staticmethod
    def _corner(small, big):
        return big[tuple(slice(0, k) for k in small.shape)]

    def _twin_load(self):
        """Copy this model's parameters and BN buffers into the leading channels of the twin."""
        tw = self._twin()
        tw._ensure_flat()
        tw.train(self.training)
        tw.two_stream_backward, tw.fold_eval_bn, tw.fast_eval = self.two_stream_backward, self.fold_eval_bn, self.fast_eval
        tw.eval_per_image, tw.defer_wgrad_levels = self.eval_per_image, self.defer_wgrad_levels
        # data parallel: the twin's engine exchanges ITS statistics and ITS flat gradient buffer (the padding channels carry
        # zeros on every rank); this model's gradients are then corners of already all-reduced tensors
        tw.grad_sync, tw.sync_bn = self.grad_sync, self.sync_bn
        src = list(self.named_parameters()) + list(self.named_buffers())
        # nothing to do when the source values are the ones already loaded (eval sweeps, repeated forwards)
        key = self._twin_src_key(tw)
        if self.__dict__.get("_twin_key") == key:
            return tw
        with torch.no_grad():
            for (_, a), (_, b) in zip(src, list(tw.named_parameters()) + list(tw.named_buffers())):
                self._corner(a, b).copy_(a) if a.dim() else b.copy_(a)
        tw.invalidate_packed()
        _lib.bump_param_generation(tw._flat_param.data_ptr())      # the twin's own caches only, not every model's
        self.__dict__["_twin_key"] = key
        return tw

    def _twin_src_key(self, tw):
        """Identity of this model's parameter / buffer values as the twin last saw them: autograd versions + the raw-pointer
        generations of every tensor (optimizers and other writers that go through `.data` / raw pointers, which autograd's
        counters do not see, bump the generation of the pointers they wrote: _lib.bump_param_generation) + the global one
        (broadcast).  A writer that does neither -- `p.data.clamp_()` -- has to call `invalidate_twin()`."""
        ts = list(self.parameters()) + list(self.buffers())
        ptrs = [t.data_ptr() for t in ts]
        return (tuple(zip(ptrs, (t._version for t in ts))), _lib.generations(ptrs), _lib.global_generation(), id(tw))

    def invalidate_twin(self):
        """Force the next forward to re-load the zero-padded twin from this model's parameters (after a `.data` write that
        neither bumps an autograd version nor calls _lib.bump_param_generation)."""
        self.__dict__.pop("_twin_key", None)

    def _twin_store_buffers(self, tw):
        """Running statistics / num_batches_tracked updated by a training-mode forward of the twin -> this model."""
        with torch.no_grad():
            for (_, a), (_, b) in zip(self.named_buffers(), tw.named_buffers()):
                a.copy_(self._corner(a, b) if a.dim() else b)
        if "_twin_key" in self.__dict__:       # model and twin are in sync again: these copies are not a change of the source
            self.__dict__["_twin_key"] = self._twin_src_key(tw)

    def _param_list(self) -> List[nn.Parameter]:
        return list(self.parameters())

    def _flat_layout(self, params):
        """Order of the parameters inside the flat buffers: parameters() order, except that the bias of every up-convolution
        (ConvTranspose2d / the bilinear branch's conv1x1) of decoder level j sits right behind the parameters of ENCODER level
        d - 1 - j.  Why: the backward fills the flat gradient buffer from its END towards its start and the data-parallel
        buckets (resdepth_amd.dp) are contiguous ranges launched as soon as all their gradients exist -- and an up-convolution's
        bias gradient is a by-product of the BatchNorm backward of the encoder level whose skip it was added to (the per-channel
        sum of the skip gradient), i.e. it completes LATE, with that encoder level.  In parameters() order those 5 small tensors
        sat inside the first (20 MB) bucket and held it back until the end of the backward (r05 diagnostics, world 1:
        bucket 0 issued 10.0 ms into a 10.2 ms step); in completion order every bucket goes out when its last weight gradient
        is done.  Optimizers and checkpoints do not see the layout (state_dict / parameters() order is unchanged; the flat
        Adam / SGD kernels are element-wise)."""
        d = self.depth
        late = {}
        for j in range(d):
            b = getattr(self._up_of(j), "bias", None)
            if b is not None:
                late[id(b)] = d - 1 - j
        enc_last = {}                     # encoder level -> index (in `params`) of its last own parameter
        pid = {id(p): i for i, p in enumerate(params)}
        for i in range(d):
            own = [pid[id(p)] for p in self.encoder[i].parameters() if id(p) in pid]
            if own:
                enc_last[i] = max(own)
        order = []
        moved = {lvl: [pid[k] for k in late if late[k] == lvl and k in pid] for lvl in range(d)}
        skip = {i for v in moved.values() for i in v}
        for i in range(len(params)):
            if i in skip:
                continue
            order.append(i)
            for lvl, last in enc_last.items():
                if last == i:
                    order.extend(sorted(moved.get(lvl, [])))
        if sorted(order) != list(range(len(params))):      # an encoder level without parameters: keep the plain order
            order = list(range(len(params)))
        return order

    def flatten_parameters(self):
        """Re-home every parameter into one flat fp32 buffer (same values, same Parameter objects) so the
        gradient all-reduce and the fused Adam step run over one contiguous range.  `_offsets[i]` = element offset of
        parameter i (parameters() order) inside the flat buffers; the layout itself follows `_flat_layout`."""
        params = self._param_list()
        dev = params[0].device
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, device=dev, dtype=torch.float32)
        offs = [0] * len(params)
        o = 0
        with torch.no_grad():
            for i in self._flat_layout(params):
                p = params[i]
                n = p.numel()
                flat[o:o + n].copy_(p.data.reshape(-1))
                p.data = flat[o:o + n].view(p.shape)
                offs[i] = o
                o += n
        self._flat_param = flat
        self._flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self._offsets = offs
        self._pack_key = None
        return flat

    def _ensure_flat(self):
        params = self._param_list()
        ok = self._flat_param is not None and self._flat_param.device == params[0].device
        if ok:
            base = self._flat_param.data_ptr()
            for p, o in zip(params, self._offsets):
                if p.data_ptr() != base + 4 * o:
                    ok = False
                    break
        if not ok:
            self.flatten_parameters()

    def grad_view(self, index: int, shape):
        o = self._offsets[index]
        n = 1
        for s in shape:
            n *= s
        return self._flat_grad[o:o + n].view(shape)

    # ------------------------------------------------------------------------------------------
    def _packed(self):
        """GEMM-layout copies of the conv / convT weights, rebuilt whenever a parameter changed, on the second HIP stream
        (the first convolution and its BN pass do not need them and run meanwhile).  Split-bf16 mode: ONE launch over a
        device-resident item table into persistent buffers (`_pack_plan`); exact-f32 / bilinear modes: one pack call per
        layer in the order the forward uses them, each with its own ready event."""
        key = self._current_pack_key()
        if self._pack_key == key and self._pack_cache is not None:
            return self._pack_cache
        d = self.depth
        pk = _Packed()
        main = torch.cuda.current_stream()
        dev = self._flat_param.device
        if self._side_stream is None or self._side_stream.device != dev:
            self._side_stream = torch.cuda.Stream(device=dev)
        side = self._side_stream
        _lib.wait_stream(side, main)             # the parameters' last writer (optimizer step) ran on the main stream

        if self.up_mode == "transpose" and _lib.tune_get("mfma_f32") == 0:
            # split-bf16 mode: every operand of the network in ONE launch into persistent buffers (device item table)
            plan = self._pack_plan()
            with torch.cuda.stream(side):
                if _lib.products() == 3:
                    _lib.zero_(plan["wamax"])    # the layers' magnitude slots: the pack fills them, then writes the fp16 forms
                _lib.check(_lib.load().rd_pack_weights_fused(plan["items"].data_ptr(), plan["n"], plan["total"], plan["tiles"],
                                                             _lib.stream_ptr()), "pack_weights_fused")
                if self._tail_expected(True):
                    up_l = self._up_of(d - 1)
                    _, v_, _, b9_ = ops.tail_compose(up_l.weight, self.last_layer.weight, up_l.bias, forward=True)
                    v_.record_stream(main)
                    b9_.record_stream(main)
                    pk.items["tail"] = (v_, b9_)
                ev = _lib.Ev(side)
            for k, tensors in plan["buffers"].items():
                pk.items[k], pk.events[k] = tensors, ev
            if "tail" in pk.items:
                pk.events["tail"] = ev
            self._pack_cache, self._pack_key = pk, key
            return pk

        def put(k, tensors):
            for t_ in tensors:
                if t_ is not None:
                    t_.record_stream(main)       # allocated under the side stream, consumed on the main stream
            ev = _lib.Ev(side)
            pk.items[k], pk.events[k] = tensors, ev

        with torch.cuda.stream(side):
            if self._first_generic():
                put("enc_first", ops.pack_conv3x3_weight(self.encoder[0][0][0].weight))
            for i in range(1, d):
                put(("enc", i - 1), ops.pack_conv3x3_weight(self.encoder[i][0][0].weight))
            put("bott", ops.pack_conv3x3_weight(self.bottleneck[0].weight))
            for i in range(d):
                up = self._up_of(i)
                put(("dec_t", i), ops.pack_conv1x1_weight(up.weight) if self.up_mode == "bilinear"
                    else ops.pack_convt2x2_weight(up.weight))
                if i < d - 1:
                    put(("dec_c", i), ops.pack_conv3x3_weight(self.decoder[i][1][0].weight))
        self._pack_cache, self._pack_key = pk, key
        return pk

    def _current_pack_key(self):
        """Identity of the parameter values the packed operands were built from: flat buffer, its raw-pointer generation
        (FusedAdam / broadcast write through `.data`), the autograd versions, and the arithmetic mode (the split-bf16 and the
        exact-f32 kernels read different layouts of the packed buffers; only the active one is written)."""
        return (self._own_param_key(), _lib.global_generation(), _lib.tune_get("mfma_f32"), _lib.products())

    def _own_param_key(self):
        """The part of the pack key that only THIS model's parameters move: flat buffer, autograd versions, and the raw-pointer
        generations of its tensors (the flat FusedAdam / FusedSGD step bumps the flat buffer's = the first parameter's, the
        per-tensor fallback every tensor's).  The backward's "parameters changed since the forward" guard compares this part:
        another model's optimizer step, a broadcast or a change of the arithmetic mode only cause a re-pack."""
        params = self._param_list()
        return (self._flat_param.data_ptr(), _lib.generations([p.data_ptr() for p in params]), tuple(p._version for p in params))

    def _pack_plan(self):
        """Persistent packed-operand buffers of every conv3x3 / ConvTranspose2d layer + the device-resident item table of
        rd_pack_weights_fused (include/resdepth_hip.h).  Built once per flat parameter buffer."""
        plan = self.__dict__.get("_pack_plan_cache")
        if plan is not None and plan["flat"] == self._flat_param.data_ptr():
            return plan
        lib, dev, d = _lib.load(), self._flat_param.device, self.depth
        rows, buffers, begin, tbegin = [], {}, 0, 0
        n_layers = 3 * d - 1 + (1 if self._first_generic() else 0)      # d - 1 encoder convs + bottleneck + d up-convs + d - 1 decoder convs
        wamax = torch.zeros(n_layers * _lib.AMAX_WORDS, dtype=torch.int32, device=dev)    # one magnitude slot per layer

        def slot(i):
            return wamax[i * _lib.AMAX_WORDS:(i + 1) * _lib.AMAX_WORDS]

        def count(kind, cout, cin, f32):
            """-> (first piece, first tile) of this item; tile-packed layers own no pieces and vice versa"""
            nonlocal begin, tbegin
            at = (begin, tbegin)
            tiles = lib.rd_pack_item_tiles(kind, cout, cin)
            if tiles > 0:
                tbegin += tiles
            else:
                begin += lib.rd_pack_item_pieces(kind, cout, cin, f32)
            return at

        def split_ptr(buf, nrows, taps, cin):
            return buf.data_ptr() + (nrows * taps * cin * 4 + 15) // 16 * 16      # the split operand follows the fp32 layout

        def conv(key, w):
            nonlocal begin
            cout, cin = w.shape[0], w.shape[1]
            wf, wd = ops._packed_buffer(cout, 9, cin, dev), ops._packed_buffer(cin, 9, cout, dev)
            sl = slot(len(rows))
            buffers[key] = (_lib.tag(wf, sl), _lib.tag(wd, sl))
            b0, t0 = count(0, cout, cin, 0)
            rows.append([w.data_ptr(), split_ptr(wf, cout, 9, cin), split_ptr(wd, cin, 9, cout), 0, cout, cin, b0, 0, t0,
                         sl.data_ptr()])

        def convt(key, w):
            nonlocal begin
            cin, cout = w.shape[0], w.shape[1]
            wtf, wtd = ops._packed_buffer(4 * cout, 1, cin, dev), ops._packed_buffer(cin, 4, cout, dev)
            sl = slot(len(rows))
            buffers[key] = (_lib.tag(wtf, sl), _lib.tag(wtd, sl))
            f32 = 1 if cin <= 128 else 0       # short-K levels may run on the exact-f32 NT kernel (fp32 operand layout)
            b0, t0 = count(1, cout, cin, f32)
            rows.append([w.data_ptr(), split_ptr(wtf, 4 * cout, 1, cin), split_ptr(wtd, cin, 4, cout), 1, cout, cin, b0,
                         wtf.data_ptr() if f32 else 0, t0, sl.data_ptr()])

        if self._first_generic():
            conv("enc_first", self.encoder[0][0][0].weight)
        for i in range(1, d):
            conv(("enc", i - 1), self.encoder[i][0][0].weight)
        conv("bott", self.bottleneck[0].weight)
        for i in range(d):
            convt(("dec_t", i), self._up_of(i).weight)
            if i < d - 1:
                conv(("dec_c", i), self.decoder[i][1][0].weight)
        items = torch.tensor(rows, dtype=torch.int64).to(dev)
        assert len(rows) == n_layers
        plan = {"flat": self._flat_param.data_ptr(), "items": items, "n": len(rows), "total": begin, "tiles": tbegin,
                "buffers": buffers, "wamax": wamax}
        self.__dict__["_pack_plan_cache"] = plan
        return plan

    def _up_of(self, i):
        """The parameterised module of decoder level i's up-convolution: the ConvTranspose2d, or the conv1x1 behind the
        nn.Upsample of the bilinear variant (lib/UNet.py:17-24)."""
        up = self.decoder[i][0] if i < self.depth - 1 else self.decoder[i]
        return up[1] if self.up_mode == "bilinear" else up

    def _tail_forward(self, cur, up, skip):
        """The last up-convolution when the last convolution follows it directly (lib/UNet.py:218-227) and the skip is level 0's
        lazy descriptor: its 64-channel full-resolution output is not built -- T = cur . V (a 1x1 convolution to 16 channels at
        the coarse resolution) and B9 are all the last convolution needs of it (ops.conv3x3_last_fwd_tail).  -> None when this
        shape / mode keeps the two-kernel route, else {skip, t16, b9, v}."""
        if not (isinstance(skip, dict) and self._tail_expected(True)):
            return None
        # V and B9 depend on the weights only: composed with the weight pack on the second stream (UNet._packed), off the
        # forward's serial chain
        pk = self._pack_cache
        if pk is not None and "tail" in pk.items and self._pack_key == self._current_pack_key():
            v, b9 = pk.get("tail")
        else:
            _, v, _, b9 = ops.tail_compose(up.weight, self.last_layer.weight, up.bias, forward=True)
        # cur: the up-convolution's input, or the lazy descriptor of the block that produces it (BN + activation on load)
        return {"skip": skip, "t16": ops.tail_t16(cur, v), "b9": b9, "v": v}

    def _tail_expected(self, lazy_skip: bool) -> bool:
        """Will the forward take the composed tail (then the block feeding the last up-convolution need not write its
        activation: `_bn_forward(want_a=False)` hands a descriptor on)?"""
        up, ll = self._up_of(self.depth - 1), self.last_layer
        return bool(self.composed_tail and self.up_mode == "transpose" and lazy_skip and ll.weight.shape[1] in (16, 32, 64)
                    and ops.tail_available(up.weight.shape[0], up.weight.shape[1]))

    def _up_forward(self, cur, packed, up, skip):
        if self.up_mode == "bilinear":
            return ops.upsample2x_add_fwd(ops.conv1x1_fwd(cur, packed[0]), up.bias, skip)
        if isinstance(skip, dict):           # skip = act(BN(z)) recomputed inside the transposed-conv epilogue
            return ops.convt2x2_fwd_bnskip(cur, packed[0], up.bias, skip["z"], skip["mean"], skip["invstd"], skip["gamma"],
                                           skip["beta"], skip["slope"], skip["slope_dev"])
        return ops.convt2x2_fwd(cur, packed[0], up.bias, skip)

    # ------------------------------------------------------------------------------------------
    def _const(self, c, value, device):
        key = (c, value, str(device))
        cache = self.__dict__.setdefault("_const_cache", {})
        if key not in cache:
            cache[key] = torch.full((c,), float(value), device=device, dtype=torch.float32)
        return cache[key]

    def _norm_of(self, block):
        """(BatchNorm2d module | None, conv bias | None) of a conv block (lib/UNet.py:36-52: with do_BN=False the block is
        conv(bias=True) -> activation)."""
        if self.do_BN:
            return block[1], None
        return None, block[0].bias

    def _act_of(self, block, name):
        """Activation of a conv block as the fused kernels take it: a float negative slope (ReLU 0, LeakyReLU 0.01), or
        the block's own nn.PReLU().weight Parameter (lib/UNet.py:27-33; index 2 with BN, 1 without)."""
        if name == "prelu":
            return block[2 if self.do_BN else 1].weight
        return _SLOPES[name]

    @staticmethod
    def _split_slope(slope):
        if isinstance(slope, torch.Tensor):
            return 0.0, slope.detach()
        return slope, None

    def _bn_forward(self, z, bn, slope, pool, training, sums=None, bias=None, want_a=True, stats=None, want_zpool=False):
        """-> (a | skip descriptor, pooled, idx, mean, invstd, count[, zpool]).  want_a=False (pooled levels): the full-resolution
        activation is not written; the first return value is then the descriptor the transposed convolution needs to
        recompute it from z in its epilogue."""
        c = z.shape[-1]
        slope, sdev = self._split_slope(slope)
        if bn is None:
            # do_BN=False: activation(conv + bias) == the fused BN-apply kernel with mean 0, invstd 1, gamma 1, beta = bias
            mean, invstd = self._const(c, 0.0, z.device), self._const(c, 1.0, z.device)
            if not pool and not want_a:
                return ({"z": z, "mean": mean, "invstd": invstd, "gamma": invstd, "beta": bias, "slope": slope, "slope_dev": sdev},
                        None, None, mean, invstd, 1)
            a, p, idx, *zp = ops.bn_act_pool_fwd(z, mean, invstd, invstd, bias, slope, pool, sdev, want_a=want_a,
                                                 want_zpool=want_zpool)
            if a is None:
                a = {"z": z, "mean": mean, "invstd": invstd, "gamma": invstd, "beta": bias, "slope": slope, "slope_dev": sdev}
            return (a, p, idx, mean, invstd, 1, *zp)
        if training and stats is not None:       # statistics already finalised by the convolution's second launch
            mean, invstd = stats
            count = z.numel() // c
        elif training:
            if sums is None:
                sums = ops.bn_stats_partial(z)
            count = z.numel() // c
            if self.sync_bn and self.grad_sync is not None:
                count = self.grad_sync.allreduce_stats(sums, count)
            mean, invstd = ops.bn_stats_finalize(sums, count, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                                 eps=bn.eps, momentum=bn.momentum)
        else:
            mean, invstd = ops.bn_eval_stats(bn.running_mean, bn.running_var, eps=bn.eps)
            count = z.numel() // c
        if not pool and not want_a:          # nothing to compute: the consumers evaluate act(BN(z)) on load (the composed tail)
            a = {"z": z, "mean": mean, "invstd": invstd, "gamma": bn.weight, "beta": bn.bias, "slope": slope, "slope_dev": sdev}
            return (a, None, None, mean, invstd, count)
        a, p, idx, *zp = ops.bn_act_pool_fwd(z, mean, invstd, bn.weight, bn.bias, slope, pool, sdev, want_a=want_a,
                                             want_zpool=want_zpool)
        if a is None:
            a = {"z": z, "mean": mean, "invstd": invstd, "gamma": bn.weight, "beta": bn.bias, "slope": slope, "slope_dev": sdev}
        return (a, p, idx, mean, invstd, count, *zp)

    # ---- inference: eval-mode BatchNorm folded into the convolutions ---------------------------------------------------
    def _can_fold(self) -> bool:
        return self.do_BN and self.up_mode == "transpose" and _lib.tune_get("mfma_f32") == 0 and self.fold_eval_bn

    def _folded(self):
        """Per conv block with BN (all but the first convolution): forward operand with the rows scaled by
        gamma / sqrt(running_var + eps), the shift beta - running_mean * scale and the activation slope.  Rebuilt when a
        parameter or a running statistic changed (lib/UNet.py:45,66,86 in eval mode)."""
        blocks = [self.encoder[i][0] for i in range(1, self.depth)] + [self.bottleneck] + \
                 [self.decoder[i][1] for i in range(self.depth - 1)]
        acts = [self.act_fn_encoder] * (self.depth - 1) + [self.act_fn_bottleneck] + [self.act_fn_decoder] * (self.depth - 1)
        key = (self._current_pack_key(),
               tuple((b[1].running_mean._version, b[1].running_var._version) for b in blocks), self._bn_gen)
        if self.__dict__.get("_fold_key") == key:
            return self.__dict__["_fold_cache"]
        out = []
        with torch.no_grad():
            for blk, act in zip(blocks, acts):
                bn = blk[1]
                scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
                shift = (bn.bias - bn.running_mean * scale).contiguous()
                a = self._act_of(blk, act)
                slope = float(a.detach()) if isinstance(a, torch.Tensor) else a     # PReLU: one read-back per weight change
                out.append((ops.pack_conv3x3_weight_folded(blk[0].weight, scale), shift, slope))
        self.__dict__["_fold_key"], self.__dict__["_fold_cache"] = key, out
        return out

    def _conv_act(self, inp, folded, pool):
        wf, shift, slope = folded
        if pool and (inp.shape[2] % 16 != 0 or inp.shape[1] % 8 != 0):
            # levels too small for the patch kernel's pooling epilogue: pool in a second (tiny) pass over the activation
            a, _ = ops.conv3x3_fwd_act(inp, wf, shift, slope, pool=False)
            c = a.shape[-1]
            zero, one = self._const(c, 0.0, a.device), self._const(c, 1.0, a.device)
            _, p, _ = ops.bn_act_pool_fwd(a, zero, one, one, zero, 1.0, True, want_a=False)     # identity BN / activation
            return a, p
        return ops.conv3x3_fwd_act(inp, wf, shift, slope, pool=pool)

    def _engine_forward_folded(self, x):
        """Eval-mode forward with no pre-BN tensor after level 0: conv -> (+shift, act[, pool]) in one kernel per block."""
        d = self.depth
        _lib.ensure_splitk_workspace(x.device)
        pk = self._packed()
        fold = self._folded()
        blk = self.encoder[0][0]
        bn = blk[1]
        act0 = self._act_of(blk, self.act_fn_encoder)
        if self.fused_first_eval and not self._first_generic() and ops.conv3x3_first_fwd_act_available(x, blk[0].weight.shape[0]):
            # level 0 in ONE kernel: convolution + eval-mode BN + activation + max-pool; what leaves is the activation itself,
            # which the decoder takes as a skip through the identity case (scale 1, shift 0, slope 1) of the descriptor it
            # evaluates on load: value-exact -- fma(a0, 1, 0) returns a0 for every a0 except that -0.0 becomes +0.0, which no
            # consumer distinguishes (the skip is added to the up-convolution's output).  Only the folded eval path takes this
            # kernel; eval forwards with fold_eval_bn = False or in exact-f32 mode write z0 and run the separate BN pass
            mean, invstd = ops.bn_eval_stats(bn.running_mean, bn.running_var, eps=bn.eps)
            slope0, sdev0 = self._split_slope(act0)
            a0, cur = ops.conv3x3_first_fwd_act(x, blk[0].weight, mean, invstd, bn.weight, bn.bias, slope0, sdev0, pool=True)
            c0 = a0.shape[-1]
            one, zero = self._const(c0, 1.0, x.device), self._const(c0, 0.0, x.device)
            skip0 = {"z": a0, "mean": zero, "invstd": one, "gamma": one, "beta": zero, "slope": 1.0, "slope_dev": None}
        else:
            if self._first_generic():
                z0 = ops.conv3x3_fwd(ops.nchw_to_nhwc(x), pk.get("enc_first")[0])
            else:
                z0 = ops.conv3x3_first_fwd(x, blk[0].weight)
            skip0, cur, _, _, _, _ = self._bn_forward(z0, bn, act0, True, False, want_a=False)
        skips = [skip0]
        for i in range(1, d):
            a, cur = self._conv_act(cur, fold[i - 1], True)
            skips.append(a)
        cur, _ = self._conv_act(cur, fold[d - 1], False)
        tail = None
        for i in range(d):
            if i == d - 1:
                tail = self._tail_forward(cur, self._up_of(i), skips[0])
                if tail is not None:
                    break
            s = self._up_forward(cur, pk.get(("dec_t", i)), self._up_of(i), skips[d - 1 - i])
            skips[d - 1 - i] = None
            cur = self._conv_act(s, fold[d + i], False)[0] if i < d - 1 else s
        x_res = x if self.do_outer_skip else None
        if self.do_outer_skip and self.do_outer_skip_BN:
            x_res, _ = self._outer_bn_forward(x, False)
        if tail is not None:
            return ops.conv3x3_last_fwd_tail(tail["skip"], tail["t16"], tail["b9"], self.last_layer.weight, self.last_layer.bias, x_res)
        return ops.conv3x3_last_fwd(cur, self.last_layer.weight, self.last_layer.bias, x_res)

    def _engine_forward(self, x, training: bool, save: bool, keep_skips: bool = False):
        """split2h mode: the pass runs under a pool of magnitude slots (_lib.AmaxPool, one torch.zeros): every kernel that writes
        a GEMM operand max-accumulates |x| into a slot of it, the tensor carries the slot as `_rd_amax`, and the GEMM that takes
        the tensor reads it -- see ops._gemm_slots.  The pool is kept with the saved activations (the weight gradients of the
        backward read the forward's slots)."""
        # Inference (eval mode, no graph kept): a per-TENSOR scale would make a tile's result depend, in the last bits, on which
        # other tiles share its batch, and the tiled sweep promises the same raster bits however the tiles are batched or sharded
        # over ranks (tests/test_blend_gpu.py).  So the folded inference path runs under a pool of per-IMAGE slot arrays
        # (rd_quant_next_img: a block scales its operand by its own image's maximum); `fast_eval` asks for per-tensor scales
        # anyway, `eval_per_image = False` for the six-product bodies of r01-r05.  Training couples the batch through BatchNorm.
        if _lib.products() != 3:
            return self._engine_forward_impl(x, training, save, keep_skips)
        if not (training or save or self.fast_eval):
            if not (self.eval_per_image and self._can_fold()):
                return self._engine_forward_impl(x, training, save, keep_skips)
            with _lib.AmaxPool(x.device, slots=32, per_image=int(x.shape[0])):
                return self._engine_forward_impl(x, training, save, keep_skips)
        pool = _lib.AmaxPool(x.device)
        with pool:
            out, S = self._engine_forward_impl(x, training, save, keep_skips)
        if S is not None:
            S["amax_pool"] = pool
        return out, S

    def _engine_forward_impl(self, x, training: bool, save: bool, keep_skips: bool = False):
        if not training and not save and self._can_fold():
            return self._engine_forward_folded(x), None
        d = self.depth
        _lib.ensure_splitk_workspace(x.device)
        pk = self._packed()
        if training:
            self._bn_gen += 1
        S = {"x": x, "enc": [], "dec": [], "training": training, "pack_key": self._pack_key} if save else None

        fused_stats = training and self.do_BN and not (self.sync_bn and self.grad_sync is not None)

        def conv_stats(inp, wf, bn):
            """-> (z, sums | None, (mean, invstd) | None).  Training: the BN statistics come out of the conv epilogue; when
            no cross-rank exchange sits between the sums and their use they are finalised right there (2 launches)."""
            if fused_stats:
                z_, mean_, invstd_ = ops.conv3x3_fwd_bn(inp, wf, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                                        eps=bn.eps, momentum=bn.momentum)
                return z_, None, (mean_, invstd_)
            if training and self.do_BN:
                return ops.conv3x3_fwd_stats(inp, wf) + (None,)
            return ops.conv3x3_fwd(inp, wf), None, None

        skips = []
        cur = None
        for i in range(d):
            blk = self.encoder[i][0]
            sums = st = None                 # training: BN statistics come out of the conv kernel's epilogue
            bn, cbias = self._norm_of(blk)
            if i == 0 and self._first_generic():
                xh = ops.nchw_to_nhwc(x)
                if save:
                    S["xh"] = xh
                z, sums, st = conv_stats(xh, pk.get("enc_first")[0], bn)
            elif i == 0:
                if fused_stats:
                    z, m_, i_ = ops.conv3x3_first_fwd_bn(x, blk[0].weight, bn.running_mean, bn.running_var,
                                                         bn.num_batches_tracked, eps=bn.eps, momentum=bn.momentum)
                    st = (m_, i_)
                elif training and self.do_BN:
                    z, sums = ops.conv3x3_first_fwd_stats(x, blk[0].weight)
                else:
                    z = ops.conv3x3_first_fwd(x, blk[0].weight)
            else:
                z, sums, st = conv_stats(cur, pk.get(("enc", i - 1))[0], bn)
            # transposed up-mode: the skip activation is recomputed from z in the decoder's convT epilogue, not stored
            lazy_skip = self.up_mode == "transpose" and not keep_skips
            want_zp = save and self.fused_bn_bwd_stats
            a, p, idx, mean, invstd, count, *zp = self._bn_forward(z, bn, self._act_of(blk, self.act_fn_encoder), True,
                                                                   training, sums, cbias, want_a=not lazy_skip, stats=st,
                                                                   want_zpool=want_zp)
            skips.append(a)
            if save:
                S["enc"].append({"z": z, "idx": idx, "mean": mean, "invstd": invstd, "count": count, "p": p,
                                 "zp": zp[0] if zp else None})
                if keep_skips:               # tests only: the backward never needs the skip activations
                    S["enc"][-1]["a"] = a
            cur = p
        bn, cbias = self._norm_of(self.bottleneck)
        zb, sums, st = conv_stats(cur, pk.get("bott")[0], bn)
        # the block feeding the LAST up-convolution need not write its activation when the composed tail runs (its two readers
        # evaluate act(BN(z)) on load): the bottleneck for depth 1, decoder block d - 2 otherwise
        lazy_last = self._tail_expected(self.up_mode == "transpose" and not keep_skips)
        ab, _, _, mean, invstd, count = self._bn_forward(zb, bn, self._act_of(self.bottleneck, self.act_fn_bottleneck),
                                                         False, training, sums, cbias, stats=st, want_a=not (lazy_last and d == 1))
        if save:
            S["bott"] = {"z": zb, "mean": mean, "invstd": invstd, "count": count, "a": ab}
        cur = ab
        tail = None
        for i in range(d):
            if i == d - 1:
                tail = self._tail_forward(cur, self._up_of(i), skips[0])
                if tail is not None:
                    if save:
                        S["dec"].append({"s": None, "tail": tail})
                    break
                assert not isinstance(cur, dict), "a lazy activation reached the two-kernel tail"
            s = self._up_forward(cur, pk.get(("dec_t", i)), self._up_of(i), skips[d - 1 - i])
            skips[d - 1 - i] = None          # the skip tensor is not needed by the backward pass
            rec = {"s": s}
            if i < d - 1:
                blk = self.decoder[i][1]
                bn, cbias = self._norm_of(blk)
                zd, sums, st = conv_stats(s, pk.get(("dec_c", i))[0], bn)
                ad, _, _, mean, invstd, count = self._bn_forward(zd, bn, self._act_of(blk, self.act_fn_decoder), False,
                                                                 training, sums, cbias, stats=st,
                                                                 want_a=not (lazy_last and i == d - 2))
                rec.update(z=zd, mean=mean, invstd=invstd, count=count, a=ad)
                cur = ad
            else:
                cur = s
            if save:
                S["dec"].append(rec)
        x_res = x if self.do_outer_skip else None
        if self.do_outer_skip and self.do_outer_skip_BN:
            x_res, obn = self._outer_bn_forward(x, training)          # BatchNorm2d(1) on channel 0 (lib/UNet.py:230-237)
            if save:
                S["outer_bn"] = obn
        if tail is not None:
            out = ops.conv3x3_last_fwd_tail(tail["skip"], tail["t16"], tail["b9"], self.last_layer.weight, self.last_layer.bias, x_res)
        else:
            out = ops.conv3x3_last_fwd(cur, self.last_layer.weight, self.last_layer.bias, x_res)
        return out, S

    def _outer_bn_forward(self, x, training):
        """BatchNorm2d(1) on x[:, 0:1].  A one-channel tensor is viewed as [P/4, 4] ("four pseudo-channels") so the
        NHWC float4 kernels apply; the four partial statistics are combined (tiny torch ops on 8 doubles)."""
        bn = self.layer_outer_skip[0]
        n, _, h, w = x.shape
        x0 = x[:, 0:1].contiguous()
        v = x0.view(1, 1, (n * h * w) // 4, 4)
        g4, b4 = bn.weight.detach().expand(4).contiguous(), bn.bias.detach().expand(4).contiguous()
        if training:
            s8 = ops.bn_stats_partial(v).view(2, 4).sum(1, keepdim=True).expand(2, 4).contiguous().view(8)
            count = n * h * w
            if self.sync_bn and self.grad_sync is not None:
                count = self.grad_sync.allreduce_stats(s8, count)
            rm4, rv4 = bn.running_mean.expand(4).contiguous(), bn.running_var.expand(4).contiguous()
            mean4, inv4 = ops.bn_stats_finalize(s8, count, rm4, rv4, bn.num_batches_tracked, eps=bn.eps,
                                                momentum=bn.momentum)
            bn.running_mean.copy_(rm4[0:1])
            bn.running_var.copy_(rv4[0:1])
        else:
            count = n * h * w
            mean4, inv4 = ops.bn_eval_stats(bn.running_mean.expand(4).contiguous(), bn.running_var.expand(4).contiguous(),
                                            eps=bn.eps)
        y, _, _ = ops.bn_act_pool_fwd(v, mean4, inv4, g4, b4, 1.0, False)       # slope 1 = identity activation
        return y.view(n, 1, h, w), {"v": v, "mean": mean4, "invstd": inv4, "g4": g4, "b4": b4, "count": count}

    def _input_grad(self, S, dz0, dout, pk):
        """dL/dx [N, Cin, T, T] (lib/UNet.py:196-246 differentiated w.r.t. its input): the first convolution's data
        gradient plus, on channel 0, the outer residual (through BatchNorm2d(1) when outer_skip_BN).  Off the training
        path (the reference never asks for it); built from existing kernels."""
        blk = self.encoder[0][0]
        w = blk[0].weight.detach()
        if self._first_generic():
            dx = ops.nhwc_to_nchw(ops.conv3x3_bwd_data(dz0, pk.get("enc_first")[1]))
        else:
            # dx[ci] = sum_{co,tap} dz[p - off(tap)][co] w[co][ci][tap] = a C -> 1 convolution of dz with the flipped taps
            planes = [ops.conv3x3_last_fwd(dz0, w[:, ci].flip(-1, -2).contiguous().unsqueeze(0), None, None)
                      for ci in range(self.n_input_channels)]
            dx = torch.cat(planes, 1) if len(planes) > 1 else planes[0]
        if self.do_outer_skip:
            if "outer_bn" in S:
                ob = S["outer_bn"]
                s12 = ob["s12"]
                tot = torch.cat([s12[0:4].sum().expand(4), s12[4:8].sum().expand(4)]).contiguous()
                g0 = ops.bn_act_bwd_apply(ob["v"], ob["mean"], ob["invstd"], ob["g4"], ob["b4"], 1.0, dout.view(ob["v"].shape),
                                          None, None, tot, ob["count"], S["training"])
                dx[:, 0:1] += g0.view(dout.shape)
            else:
                dx[:, 0:1] += dout
        return dx

    def _engine_backward(self, S, dout, want_dx=False):
        if _lib.products() != 3:
            return self._engine_backward_impl(S, dout, want_dx)
        with _lib.AmaxPool(dout.device):         # the gradient operands' magnitude slots (a fresh block per backward)
            return self._engine_backward_impl(S, dout, want_dx)

    def _engine_backward_impl(self, S, dout, want_dx=False):
        """Writes every parameter gradient into a flat gradient buffer; returns the list of views
        (state_dict / parameters() order).  Order of production: head, decoder levels d-1..0 + bottleneck,
        encoder levels d-1..0 -- i.e. from the END of the flat buffer towards its start, which is what the
        data-parallel bucketing in resdepth_amd.dp relies on for overlap."""
        d = self.depth
        if S.get("pack_key") is not None and self._own_param_key() != S["pack_key"][0]:
            # the packed operands live in persistent buffers that a later forward re-packs in place: the data gradients
            # of THIS graph would silently use the new weights.  torch raises in the same situation ("one of the variables
            # needed for gradient computation has been modified by an inplace operation")
            raise RuntimeError("resdepth_amd.UNet: a parameter was modified (optimizer step / load_state_dict / .data write) "
                               "between this graph's forward and its backward")
        _lib.ensure_splitk_workspace(dout.device)
        pk = self._packed()
        training = S["training"]
        params = self._param_list()
        index = {id(p): i for i, p in enumerate(params)}
        grads = [None] * len(params)
        # autograd ACCUMULATES into an existing .grad; if the caller kept gradients from a previous step
        # (the reference sets them to None, lib/Trainer.py:221-222) they may alias the persistent flat
        # buffer, so write this step's gradients somewhere else.
        flat = self._flat_grad
        if any(p.grad is not None for p in params):
            flat = torch.empty_like(self._flat_grad)
        sync = self.grad_sync if flat is self._flat_grad else None
        sync_bn = self.sync_bn and self.grad_sync is not None and training

        def gv(p):
            i = index[id(p)]
            o = self._offsets[i]
            g = flat[o:o + p.numel()].view(p.shape)
            grads[i] = g
            return g

        # Two-stream backward: the weight gradients are off the critical path (dz -> dgrad -> BN-bwd -> dz' ...) and the
        # HBM-bound elementwise kernels on that path complement the MFMA-bound wgrad GEMMs, so every wgrad is enqueued
        # on a second HIP stream behind an event recorded after its dz.  Tensors produced on the main stream and read
        # there are pinned with record_stream; the main stream joins the side stream before the backward returns.
        main = torch.cuda.current_stream()
        side = None
        if self.two_stream_backward:
            if self._side_stream is None or self._side_stream.device != dout.device:
                self._side_stream = torch.cuda.Stream(device=dout.device)
            side = self._side_stream
            if sync is not None:
                sync.launch_stream = side
        elif sync is not None:
            sync.launch_stream = None

        def done(*ps):
            if sync is not None:
                sync.params_ready(self, [index[id(p)] for p in ps if p is not None])

        def wgrad(fn, reads, *args, ready=()):
            """run fn(*args) (a weight-gradient op writing into the flat gradient buffer) on the side stream"""
            if side is None:
                fn(*args)
                done(*ready)
                return
            ev = _lib.Ev(main)
            with torch.cuda.stream(side):
                _lib.ev_wait(side, ev)
                fn(*args, ws_slot=1)
                for t_ in reads:
                    t_.record_stream(side)
                done(*ready)

        fused_stats = self.fused_bn_bwd_stats

        def hook(rec, block, act_name, mode=1):
            """ops.BnHook of a conv block (its saved z / statistics / affine parameters): handed to the kernel that
            produces the gradient w.r.t. the block's activation, whose epilogue then emits the BN-backward sums."""
            if not fused_stats or (mode == 2 and rec.get("zp") is None):
                return None
            bn, cbias = self._norm_of(block)
            slope, sdev = self._split_slope(self._act_of(block, act_name))
            z = rec["z"] if mode == 1 else rec["zp"]
            if bn is None:
                return ops.BnHook(z, rec["mean"], rec["invstd"], self._const(z.shape[-1], 1.0, z.device), cbias, slope, sdev, mode)
            return ops.BnHook(z, rec["mean"], rec["invstd"], bn.weight, bn.bias, slope, sdev, mode)

        def with_stats(fn, *args, bn=None):
            """fn(*args[, bn=hook]) -> (tensor, (partial rows, count) | None)"""
            if bn is None:
                return fn(*args), None
            out, part = fn(*args, bn=bn)
            return out, (part if part[1] > 0 else None)

        def bn_backward(rec, block, act_name, g_full, g_pool, idx, extra_bias=None, pre=(), fuse_into=None):
            """-> dz of the block.  fuse_into (level 0 only): a weight-gradient op that evaluates dz itself from the arguments
            of the apply pass (ops.conv3x3_first_bwd_weight_bn) -- the statistics are reduced here, dz is never written and
            None is returned."""
            c = rec["z"].shape[-1]
            bn, cbias = self._norm_of(block)
            act = self._act_of(block, act_name)
            slope, sdev = self._split_slope(act)
            prelu_w = act if sdev is not None else None
            # statistics that came out of the producers' epilogues (one per gradient operand); all must be there
            operands = (g_full is not None) + (g_pool is not None)
            pre = [q for q in pre if q is not None]
            pre = pre if len(pre) == operands else None
            if isinstance(g_full, ops.LastConvGrad) and (fuse_into is None or not pre):
                g_full = g_full.tensor()         # a consumer that needs the tensor after all

            dextra = gv(extra_bias) if extra_bias is not None else None     # ConvTranspose2d bias (skip add): by-product

            def side_grads(sums):
                # the other by-product of the same reduction: the PReLU slope
                if prelu_w is not None:
                    gv(prelu_w).copy_(sums[3 * c:4 * c].sum().reshape(1))

            if bn is None:
                # do_BN=False: a = act(z + bias); d bias = sum g', dz = g' (the "eval" form of the fused kernels)
                one = self._const(c, 1.0, rec["z"].device)
                if pre:
                    sums = ops.bn_bwd_stats_finalize(pre, c, dextra=dextra)
                else:
                    sums = ops.bn_act_bwd_reduce(rec["z"], rec["mean"], rec["invstd"], one, cbias, slope, g_full, g_pool, idx,
                                                 slope_dev=sdev, dextra=dextra)
                side_grads(sums)
                dz = ops.bn_act_bwd_apply(rec["z"], rec["mean"], rec["invstd"], one, cbias, slope, g_full, g_pool, idx,
                                          sums, 1.0, False, dgamma=None, dbeta=gv(cbias), slope_dev=sdev)
                done(cbias, prelu_w, extra_bias)
                return dz
            if pre:
                sums = ops.bn_bwd_stats_finalize(pre, c, dgamma=None if sync_bn else gv(bn.weight),
                                                 dbeta=None if sync_bn else gv(bn.bias), dextra=dextra)
            elif sync_bn:
                sums = ops.bn_act_bwd_reduce(rec["z"], rec["mean"], rec["invstd"], bn.weight, bn.bias, slope, g_full,
                                             g_pool, idx, slope_dev=sdev, dextra=dextra)
            else:       # the reduction writes dgamma / dbeta itself (no separate launch)
                sums = ops.bn_act_bwd_reduce(rec["z"], rec["mean"], rec["invstd"], bn.weight, bn.bias, slope, g_full,
                                             g_pool, idx, slope_dev=sdev, dgamma=gv(bn.weight), dbeta=gv(bn.bias),
                                             dextra=dextra)
            side_grads(sums)
            if sync_bn:
                local = sums[:2 * c].clone()
                self.grad_sync.allreduce_sums(sums)
                gv(bn.weight).copy_(local[c:2 * c])
                gv(bn.bias).copy_(local[:c])
            if fuse_into is not None:
                done(bn.weight, bn.bias, prelu_w, extra_bias)
                fuse_into(rec["z"], rec["mean"], rec["invstd"], bn.weight, bn.bias, slope, g_full, g_pool, idx, sums, rec["count"],
                          training, sdev)
                return None
            dz = ops.bn_act_bwd_apply(rec["z"], rec["mean"], rec["invstd"], bn.weight, bn.bias, slope, g_full,
                                      g_pool, idx, sums, rec["count"], training, slope_dev=sdev)
            done(bn.weight, bn.bias, prelu_w, extra_bias)
            return dz

        dout = dout.contiguous()
        c0 = self.filter_depths[0]
        # head (lib/UNet.py:227-244): the outer residual add passes dout straight through to x (not needed)
        ll = self.last_layer
        fwd_tail = S["dec"][d - 1].get("tail")      # the forward never built the last convolution's input (_tail_forward)
        if fwd_tail is None:
            wgrad(lambda *a, **k: ops.conv3x3_last_bwd_weight(*a, want_bias=ll.bias is not None, **k), (dout,),
                  S["dec"][d - 1]["s"], dout, gv(ll.weight), gv(ll.bias) if ll.bias is not None else None,
                  ready=(ll.weight, ll.bias))
        if "outer_bn" in S:
            # BatchNorm2d(1) on the outer skip: only its affine parameters need gradients (x is an input)
            ob = S["outer_bn"]
            obn = self.layer_outer_skip[0]
            s12 = ops.bn_act_bwd_reduce(ob["v"], ob["mean"], ob["invstd"], ob["g4"], ob["b4"], 1.0,
                                        dout.view(ob["v"].shape), None, None)
            ob["s12"] = s12
            gv(obn.bias).copy_(s12[0:4].sum().reshape(1))
            gv(obn.weight).copy_(s12[4:8].sum().reshape(1))
            done(obn.weight, obn.bias)
        # every data-gradient kernel below also emits the BN-backward sums of the block that consumes its output
        # (`hook`): the skip gradient of encoder level j (full part), the pooled gradient (pooled part), the decoder /
        # bottleneck block behind a transposed convolution
        hook0 = hook(S["enc"][0], self.encoder[0][0], self.act_fn_encoder)
        up_last = self._up_of(d - 1)
        tail = fwd_tail is not None or (self.composed_tail and self.up_mode == "transpose"
                                        and ops.tail_available(up_last.weight.shape[0], up_last.weight.shape[1]))
        fuse_first = (self.fused_first_wgrad and self.do_BN and not want_dx and not self._first_generic()
                      and ops.conv3x3_first_bwd_weight_bn_available(S["x"], c0))
        if tail and fuse_first and hook0 is not None and c0 in (16, 32, 64):
            # g = conv_last^T(dout), C0 channels at full resolution, has three readers -- the last up-convolution's two gradients
            # and level 0's BN backward -- and all three evaluate what they need from the 1-channel dout: only the statistics of
            # the hook leave this launch, the tensor is never written
            wpart = None
            if fwd_tail is not None:
                # ... and with the forward tail the last convolution's weight gradient needs the same two operands (z of level
                # 0, the dout tile): one pass gives its partial sums and the hook's statistics
                wpart, st = ops.conv3x3_last_bwd_tail_fused(fwd_tail["skip"], dout, ll.weight)
            else:
                _, st = ops.conv3x3_last_bwd_data(dout, ll.weight, c0, bn=hook0, write=False)
            st = st if st[1] > 0 else None
            g = ops.LastConvGrad(dout, ll.weight, c0)
        else:
            wpart = None
            g, st = with_stats(ops.conv3x3_last_bwd_data, dout, ll.weight, c0, bn=hook0)
        skipgrad, skipstat = [None] * d, [None] * d
        gp = gpstat = None
        for i in reversed(range(d)):
            up = self._up_of(i)
            src = S["dec"][i - 1] if i > 0 else S["bott"]
            if self.up_mode == "bilinear":
                dt = ops.upsample2x_bwd(g)            # adjoint of the interpolation; then the coarse-grid conv1x1
                wgrad(ops.conv1x1_bwd_weight, (dt,), src["a"], dt, gv(up.weight), ready=(up.weight,))
                dprev, dstat = ops.conv1x1_bwd_data(dt, pk.get(("dec_t", i))[1]), None
            else:
                sblk, sact = (self.decoder[i - 1][1], self.act_fn_decoder) if i > 0 else (self.bottleneck, self.act_fn_bottleneck)
                if i == d - 1 and tail:
                    # the last up-convolution feeds the last convolution directly (lib/UNet.py:218-227): its two gradients are
                    # stencils / correlations on the 1-channel dout, the C0-channel gradient g is not an operand (ops.tail_*)
                    if fwd_tail is not None:
                        tail_v = fwd_tail["v"]
                        c16 = torch.empty(up.weight.shape[0], 16, device=dout.device, dtype=torch.float64)
                        wgrad(lambda *a, **k: ops.convt_last_bwd_weight(*a, c16=c16, **k), (dout, tail_v, c16), src["a"], dout,
                              ll.weight, gv(up.weight), ready=(up.weight,))
                        # ... and the last convolution's own weight gradient, whose input s was never a tensor: z of level 0 +
                        # dout, the correlations c16 just computed, the up-convolution's weight and bias
                        if wpart is not None:      # its pass over z already ran with the statistics hook (head of the backward)
                            wgrad(lambda *a, **k: ops.tail_wl_finish(*a, want_bias=ll.bias is not None, **k), (wpart,), wpart, c16,
                                  up.weight, up.bias, gv(ll.weight), gv(ll.bias) if ll.bias is not None else None,
                                  ready=(ll.weight, ll.bias))
                        else:
                            wgrad(lambda *a, **k: ops.conv3x3_last_bwd_weight_tail(*a, want_bias=ll.bias is not None, **k), (dout,),
                                  fwd_tail["skip"], dout, c16, up.weight, up.bias, gv(ll.weight),
                                  gv(ll.bias) if ll.bias is not None else None, ready=(ll.weight, ll.bias))
                    else:
                        _, tail_v = ops.tail_compose(up.weight, ll.weight)
                        wgrad(ops.convt_last_bwd_weight, (dout, tail_v), src["a"], dout, ll.weight, gv(up.weight), ready=(up.weight,))
                    dprev, dstat = with_stats(ops.convt_last_bwd_data, dout, tail_v, bn=hook(src, sblk, sact))
                else:
                    assert not isinstance(g, ops.LastConvGrad)
                    wgrad(ops.convt2x2_bwd_weight, (g,), src["a"], g, gv(up.weight), ready=(up.weight,))
                    dprev, dstat = with_stats(ops.convt2x2_bwd_data, g, pk.get(("dec_t", i))[1], bn=hook(src, sblk, sact))
            skipgrad[d - 1 - i], skipstat[d - 1 - i] = g, st       # gradient wrt the encoder skip a_{d-1-i} (SkipConnection is an ADD)
            if i > 0:
                blk = self.decoder[i - 1][1]
                dz = bn_backward(src, blk, self.act_fn_decoder, dprev, None, None, pre=(dstat,))
                wgrad(ops.conv3x3_bwd_weight, (dz,), S["dec"][i - 1]["s"], dz, gv(blk[0].weight), ready=(blk[0].weight,))
                j = d - i                 # encoder level whose skip a_j was added to this decoder level's up-convolution
                g, st = with_stats(ops.conv3x3_bwd_data, dz, pk.get(("dec_c", i - 1))[1],
                                   bn=hook(S["enc"][j], self.encoder[j][0], self.act_fn_encoder))
            else:
                dz = bn_backward(src, self.bottleneck, self.act_fn_bottleneck, dprev, None, None, pre=(dstat,))
                wgrad(ops.conv3x3_bwd_weight, (dz,), S["enc"][d - 1]["p"], dz, gv(self.bottleneck[0].weight),
                      ready=(self.bottleneck[0].weight,))
                gp, gpstat = with_stats(ops.conv3x3_bwd_data, dz, pk.get("bott")[1],
                                        bn=hook(S["enc"][d - 1], self.encoder[d - 1][0], self.act_fn_encoder, mode=2))
        for i in reversed(range(d)):
            e = S["enc"][i]
            blk = self.encoder[i][0]
            # third reduction output = per-channel sum of the skip gradient = bias gradient of the
            # ConvTranspose2d whose output was added to this skip (decoder level d-1-i)
            j = d - 1 - i
            up = self._up_of(j)
            fuse = None
            if i == 0 and fuse_first:
                # level 0: dz has one reader, the first convolution's weight gradient -- which evaluates it itself.  On the MAIN
                # stream, where the apply pass it replaces ran: behind the side stream's queue of strip kernels it lengthens the
                # tail of the step (interleaved: -0.4 % there, +1.1 % here, against the two-kernel route)
                def fuse(z_, mean_, invstd_, gamma_, beta_, slope_, gf_, gp_, idx_, sums_, count_, training_, sdev_, w_=blk[0].weight):
                    ops.conv3x3_first_bwd_weight_bn(S["x"], z_, mean_, invstd_, gamma_, beta_, slope_, gf_, gp_, idx_, sums_, count_,
                                                    training=training_, slope_dev=sdev_, out=gv(w_))
                    if side is None or sync is None:     # (the event below only orders a bucket launch: nothing to order without one)
                        done(w_)
                        return
                    ev = _lib.Ev(main)
                    with torch.cuda.stream(side):      # bucket launches are ordered on the side stream
                        _lib.ev_wait(side, ev)
                        done(w_)
            dz = bn_backward(e, blk, self.act_fn_encoder, skipgrad[i], gp, e["idx"], extra_bias=up.bias,
                             pre=(skipstat[i], gpstat), fuse_into=fuse)
            skipgrad[i] = skipstat[i] = None
            if dz is None:
                continue
            if i > 0:
                # The last `defer_wgrad_levels` encoder levels launch their weight gradient BEHIND their data gradient (the side
                # stream's event is recorded after it): the data gradient then has the chip to itself -- it is the critical path --
                # and the strip kernel runs beside what follows on the main stream, which at level 1 is the fused first-convolution
                # weight gradient: an HBM / VALU-class kernel that otherwise ends the step alone (profiles/r06_notes.md section 12).
                # Two MFMA-class kernels side by side only share one power budget; an MFMA-class one beside an HBM-class one overlaps.
                late = side is not None and i <= self.defer_wgrad_levels
                if not late:
                    wgrad(ops.conv3x3_bwd_weight, (dz,), S["enc"][i - 1]["p"], dz, gv(blk[0].weight), ready=(blk[0].weight,))
                gp, gpstat = with_stats(ops.conv3x3_bwd_data, dz, pk.get(("enc", i - 1))[1],
                                        bn=hook(S["enc"][i - 1], self.encoder[i - 1][0], self.act_fn_encoder, mode=2))
                if late:
                    wgrad(ops.conv3x3_bwd_weight, (dz,), S["enc"][i - 1]["p"], dz, gv(blk[0].weight), ready=(blk[0].weight,))
            elif self._first_generic():
                wgrad(ops.conv3x3_bwd_weight, (dz,), S["xh"], dz, gv(blk[0].weight), ready=(blk[0].weight,))
            else:
                wgrad(ops.conv3x3_first_bwd_weight, (dz,), S["x"], dz, gv(blk[0].weight), ready=(blk[0].weight,))
        dx = self._input_grad(S, dz, dout, pk) if want_dx else None
        if side is not None:
            _lib.wait_stream(main, side)     # every weight gradient (and bucket launch) is ordered before what follows
        if sync is not None:
            sync.finish(self)
        elif self.grad_sync is not None:
            self.grad_sync.allreduce_unbucketed(flat)
        return grads, dx

    # ------------------------------------------------------------------------------------------
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("resdepth_amd.UNet runs on MI355X only: move the model and the input to a HIP device "
                               "(there is no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != self.n_input_channels:
            raise ValueError(f"expected input [N, {self.n_input_channels}, T, T], got {tuple(x.shape)}")
        th, tw = x.shape[2], x.shape[3]
        q = 2 ** self.depth
        if th < q or tw < q or th % q or tw % q:
            # the reference fails later, in the first skip ADD whose pooled / up-sampled sizes disagree (lib/UNet.py:219)
            raise ValueError(f"tile height and width must be multiples of 2^depth = {q} (got {tuple(x.shape[2:])})")
        if self.training and self.do_BN and x.shape[0] * (th // q) * (tw // q) == 1:
            # torch.nn.functional.batch_norm's own check, hit by the reference at its bottleneck BatchNorm2d (lib/UNet.py:66)
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([1, {self.filter_depths[-1]}, 1, 1])")
        x = x.contiguous().float()
        with _lib.device_of(x):          # raw kernel launches go to the CURRENT device's stream: make it x's device
            params = self._param_list()
            if params[0].device != x.device:
                raise RuntimeError(f"resdepth_amd.UNet: input on {x.device} but parameters on {params[0].device}")
            need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
            if self._needs_twin():
                tw = self._twin_load()
                if not need_grad:
                    out, _ = tw._engine_forward(self._pad_input(x, tw), tw.training, save=False)
                    if self.training:
                        self._twin_store_buffers(tw)
                    return out
                return _TwinFunction.apply(x, self, tw, *params)
            self._ensure_flat()
            if not need_grad:
                out, _ = self._engine_forward(x, self.training, save=False)
                return out
            return _UNetFunction.apply(x, self, *params)

    @staticmethod
    def _pad_input(x, tw):
        extra = tw.n_input_channels - x.shape[1]
        if extra == 0:
            return x
        return torch.cat([x, x.new_zeros(x.shape[0], extra, x.shape[2], x.shape[3])], 1)

    def invalidate_packed(self):
        """Drop the packed (GEMM-layout) weight copies; the next forward re-packs.  Needed only after writing parameters
        through `.data` / raw pointers without going through FusedAdam or resdepth_amd.dp (both invalidate themselves)."""
        self._pack_key = None
        self._pack_cache = None


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, *params):
        out, saved = model._engine_forward(x, model.training, save=True)
        ctx.model = model
        ctx.saved = saved
        return out

    @staticmethod
    def backward(ctx, dout):
        model, S = ctx.model, ctx.saved
        if S is None:
            raise RuntimeError("resdepth_amd.UNet: backward called twice (activations were released; set "
                               "model.retain_activations = True for loss.backward(retain_graph=True))")
        with _lib.device_of(dout):
            grads, dx = model._engine_backward(S, dout, want_dx=ctx.needs_input_grad[0])
        if not model.retain_activations:
            ctx.saved = None
        return (dx, None, *grads)


class _TwinFunction(torch.autograd.Function):
    """Forward / backward of a UNet whose channel counts are not multiples of 4, executed on its zero-padded twin
    (UNet._twin): gradients are the leading-channel corners of the twin's."""

    @staticmethod
    def forward(ctx, x, model, twin, *params):
        out, saved = twin._engine_forward(UNet._pad_input(x, twin), twin.training, save=True)
        if model.training:
            model._twin_store_buffers(twin)
        ctx.model, ctx.twin, ctx.saved, ctx.cin = model, twin, saved, x.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        model, twin, S = ctx.model, ctx.twin, ctx.saved
        if S is None:
            raise RuntimeError("resdepth_amd.UNet: backward called twice (activations were released)")
        with _lib.device_of(dout):
            grads, dx = twin._engine_backward(S, dout, want_dx=ctx.needs_input_grad[0])
            out = [UNet._corner(p, g).clone() if p.requires_grad else None for p, g in zip(model._param_list(), grads)]
            if dx is not None:
                dx = dx[:, :ctx.cin].contiguous()
        if not model.retain_activations:
            ctx.saved = None
        return (dx, None, None, *out)
