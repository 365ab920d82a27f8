"""Key-net assembly and the top of the hot path (mirror of keynet/system.py:26-516).

    (sensor, model) = PermutationKeynet(inshape, net)          # factories, keynet/system.py:472-510
    y = model.forward(sensor.fromtensor(x).encrypt().astensor())   # keynet/system.py:130-133, README.md:27-34

Keying (host, offline) follows the reference's algebra call for call so that, under the same numpy seed, the stored
operators are identical (tests/test_host_keying.py compares CSR triplets with the golden vectors).  The forward runs on
the MI355X: layers exchange feature-major [D+1, N] blocks, ReLU is fused into the producing kernel, and batches are
supported (the reference's KeyedModel.forward is N=1 only).
"""
import copy
import os
import warnings
from collections import OrderedDict
import numpy as np
import scipy.sparse
import torch
from torch import nn

from . import torch as ktorch
from . import sparse as ksp
from . import layer as klayer
from . import _capi
from .globals import verbose
from .util import find_closest_positive_divisor
from .sparse import sparse_identity_matrix
from .keys import keygen, diagonal_affine_to_linear   # noqa: F401  (re-exported: the reference exposes keygen from keynet.system)


class KeyedModel(object):
    """The keyed network: an nn.Sequential of KeyedLayer / nn.ReLU plus the two end keys (keynet/system.py:26-157)."""

    def __init__(self, net, inshape, inkey, f_layername_to_keypair, f_module_to_keyedmodule=None, do_output_encryption=False):
        """net: source nn.Module with uniquely named children ('reluN' after a linear layer, '<layer>_bn' batch norms,
        'dropoutN'); inkey: decryption key of the sensor (= Ainv of the first layer); f_layername_to_keypair(name, outshape)
        -> (A, Ainv) draws one key pair per traced layer, in trace order; f_module_to_keyedmodule is the layergen seam."""
        net.eval()
        chain = self._trace(net, inshape)
        (keys, last) = self._draw_keys(chain, inkey, f_layername_to_keypair, do_output_encryption)
        self._keynet = nn.Sequential(self._key_layers(net, chain, keys, f_module_to_keyedmodule))
        self._embeddingkey = keys[last]['outinv'] if do_output_encryption else None
        self._imagekey = inkey
        self._layernames = set(name for (name, _) in net.named_children())
        self._outshape = chain['output']['outshape']

    @staticmethod
    def _trace(net, inshape):
        """Shape trace with the identity layers spliced out of the prev/next links.  The dropout ENTRIES stay in the table
        (the reference's filter at keynet/system.py:35 tests whether the layer name is a substring of 'dropout', which
        real names such as 'dropout0' are not), so each of them still draws a key pair below: part of RNG parity."""
        chain = ktorch.netshape(net, inshape)
        marker = 'dropout'
        chain = OrderedDict((k, v) for (k, v) in chain.items() if k not in marker)
        for v in chain.values():
            if v['nextlayer'] is not None and marker in v['nextlayer']:
                v['nextlayer'] = chain[v['nextlayer']]['nextlayer']
            elif v['prevlayer'] is not None and marker in v['prevlayer']:
                v['prevlayer'] = chain[v['prevlayer']]['prevlayer']
        return chain

    @staticmethod
    def _draw_keys(chain, inkey, draw, do_output_encryption):
        """name -> {'A': output key (None on the last layer unless the output is encrypted), 'Ainv': inverse of the key on
        the layer's input, 'outinv': inverse of its own output key}.  One draw per traced layer, in trace order."""
        last = chain['output']['prevlayer']
        pairs = OrderedDict((k, draw(k, v['outshape'])) for (k, v) in chain.items() if k not in ('input', 'output'))
        keys = {}
        for (k, (A, Ainv)) in pairs.items():
            prev = chain[k]['prevlayer']
            keys[k] = {'A': A if (k != last or do_output_encryption) else None, 'Ainv': inkey if prev == 'input' else pairs[prev][1], 'outinv': Ainv}
        return (keys, last)

    @staticmethod
    def _key_layers(net, chain, keys, make):
        """Walk the children and emit the keyed sequence.  A linear layer followed by a ReLU (or a '<name>_bn' batch norm)
        is keyed together with it, using the follower's output key; Dropout vanishes."""
        out = OrderedDict()

        def rekeyed(follower, target):
            # output key of `target` as seen through `follower`:  (A_f . A_f_in^-1) . A_target
            return keys[follower]['A'].dot(keys[follower]['Ainv']).dot(keys[target]['A'])

        for (name, m) in net.named_children():
            if verbose():
                print('[keynet_amd.KeyedModel]: keying "%s"' % name)
            assert name in keys and name in chain, 'layer "%s" was not reached by the shape trace' % name
            node = chain[name]
            if isinstance(m, nn.Dropout):
                continue
            if isinstance(m, nn.BatchNorm2d):
                host = name.split('_')[0]
                assert '_bn' in name and node['prevlayer'] == host, 'a batch norm must be named "<layer>_bn" and follow "<layer>" directly'
                fused = copy.deepcopy(getattr(net, host))
                (w, b) = ktorch.fuse_conv2d_and_bn(fused.weight, fused.bias, m.running_mean, m.running_var, 1E-5, m.weight, m.bias)
                (fused.weight, fused.bias) = (torch.nn.Parameter(w), torch.nn.Parameter(b))
                out[host] = make(fused, chain[host]['inshape'], node['outshape'], rekeyed(name, host), keys[host]['Ainv'])
            elif isinstance(m, nn.ReLU):
                host = node['prevlayer']
                if '_bn' in host:
                    warnings.warn('ReLU right after the batch norm "%s": it has to be keyed on its own (costly); avoid bn -> relu chains' % host)
                    out[name] = make(m, node['inshape'], node['outshape'], keys[name]['A'], keys[name]['Ainv'])
                else:
                    out[host] = make(getattr(net, host), chain[host]['inshape'], chain[host]['outshape'], rekeyed(name, host), keys[host]['Ainv'])
                    out[name] = copy.deepcopy(m)          # stays a plain ReLU
            else:
                nxt = node['nextlayer']
                if nxt is not None and (nxt == '%s_bn' % name or 'relu' in nxt):
                    continue                               # emitted when its follower is reached
                out[name] = make(m, node['inshape'], node['outshape'], keys[name]['A'], keys[name]['Ainv'])
        return out

    @classmethod
    def fromlayers(cls, layers, outshape, imagekey=None, embeddingkey=None):
        """Assemble from already keyed layers (OrderedDict name -> KeyedLayer | nn.ReLU): public key-nets, fixtures."""
        self = cls.__new__(cls)
        self._keynet = nn.Sequential(OrderedDict(layers))
        (self._embeddingkey, self._imagekey, self._layernames, self._outshape) = (embeddingkey, imagekey, set(layers.keys()), outshape)
        return self

    def __repr__(self):
        return self._keynet.__repr__()

    def __getstate__(self):
        """Whole key-nets are pickled by the reference's users (test/test_keynet.py:106, vipy.util.save): device-side state -- the overlapped
        forward's workspaces and streams, the whole-net kernel's handle -- is dropped and rebuilt on first use (operators: SparseMatrix.__getstate__)."""
        d = dict(self.__dict__)
        d.pop('_overlap_plans', None)
        d.pop('_chain_ops', None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)

    def __getattr__(self, attr):
        if attr.startswith('__') or '_keynet' not in self.__dict__:
            raise AttributeError(attr)
        return getattr(self.__dict__['_keynet'], attr)

    # -- the hot path -------------------------------------------------------------------------------------------
    RESCREEN = True              # re-screen the float-key contract on every forward (KN_NO_RESCREEN=1: A/B switch, read per call)
    RESCREEN_MAX_PASSES = 4

    def forward_linear(self, img_cipher, overlap=None, _slots_out=None):
        """[N, D0+1] -> [N, classes+1]: the nn.Sequential of keynet/system.py:132 with the unkeyed ReLUs fused into the
        producing layer's kernel epilogue.  Stream-ordered on torch's current HIP stream.  Host synchronisation: none for key-nets whose
        layers all run under a DECLARED contract (exact=True: the permutation key-nets; exact=False: forced); a key-net with layers on the
        matrix cores by a CALIBRATION decision (exact='auto': the float-key contract) reads one float per keyed layer back at the end of
        every forward -- max |x| of each such layer, gathered on the device inside the producing kernels (kn_spmm_screen) -- and, when a
        layer's input has outgrown its calibration by more than KeyedLayer.RESCREEN_FACTOR, re-calibrates that layer on this batch and runs
        the batch again.  What that checks (and no more): the batch maximum of each screened layer's input against the maximum the
        decision was calibrated on, with the 2x headroom every accepted decision has (element-wise |d| <= 1e-5 + 1e-5 |ref|, measured on up to 256
        columns of the calibration batch; an unmeasured dense layer is accepted only with its worst-case bound at half the tolerance); NaN activations
        are not screened.  The reference applies one arithmetic on every call (keynet/sparse.py:488-492): only the 'exact' contract IS that arithmetic.
        The first forward of an 'auto' layer calibrates it (host reads).
        `overlap`: run the batch as two half-batch column windows on two side streams, one kernel apart (see _forward_overlapped);
        None = automatically for device-resident feature-major batches that are a multiple of 256 images, False = never.
        Memory: the overlapped forward keeps two ping-pong workspaces of max_rows x N floats per (device, N) plan (VGG-16 at N = 256:
        2 x 3.3 GB) plus two side streams; at most OVERLAP_PLANS_KEPT plans are cached (least recently used dropped),
        release_workspace() drops them all."""
        if not img_cipher.is_cuda and img_cipher.dim() == 2 and torch.cuda.is_available() and _slots_out is None:
            # a host tensor (how the reference's users call it): ONE copy to the device here instead of one per layer, so that the layers
            # chain on the device and the per-forward contract screen sees them; the result goes back where the input lives
            return self.forward_linear(img_cipher.detach().float().cuda(), overlap=overlap).to(img_cipher.device)
        keyed = [c for c in self._keynet.children() if isinstance(c, klayer.KeyedLayer)]
        on_dev = img_cipher.is_cuda and img_cipher.dim() == 2
        capturing = on_dev and torch.cuda.is_current_stream_capturing()
        if on_dev and not capturing and _slots_out is None and img_cipher.dtype == torch.float32 and self._has_tiled_conv(keyed):
            # The fast loaders address an activation block with 32-bit element offsets: a layer of R rows takes them while R x N < 2^31.  VGG-16's conv1_2 (3.2 M rows) leaves
            # them at 1 024 images -- 113.7 ms in the stored order against 2 x 31 ms for two passes of 512 (profiles/r06_vgg16_other_batches.txt) -- so a batch that large runs
            # as passes of the largest multiple of 256 images that keeps every layer inside.
            big = max(max(c.W.shape) for c in keyed)
            chunk = (self.MAX_BLOCK_ELEMENTS - 1) // big // 256 * 256
            if 0 < chunk < img_cipher.shape[0]:
                n = img_cipher.shape[0]
                self._chunked_forwards = getattr(self, '_chunked_forwards', 0) + 1
                outs = []
                for lo in range(0, n, chunk):
                    part = img_cipher[lo:lo + chunk]
                    if img_cipher.t().is_contiguous():             # (a window of feature-major memory: its own feature-major block, so that a pass takes the overlapped form)
                        part = part.detach().t().contiguous().t()
                    outs.append(self.forward_linear(part, overlap=overlap))
                return torch.cat(outs, dim=0)
        if (on_dev and not capturing and _slots_out is None and img_cipher.dtype == torch.float32 and img_cipher.shape[0] % self.BATCH_TILE == 0 and
                not img_cipher.t().is_contiguous() and self._has_tiled_conv(keyed)):
            # a row-major [N, D] batch (how a caller of the reference holds it): the kernels read feature-major memory, so the first layer would copy it anyway -- done here, once,
            # the forward can also take its two-window overlapped form (VGG-16 at 256 images: 58.7 -> 57.0 ms, tools/layout_time.py)
            img_cipher = img_cipher.detach().t().contiguous().t()
        if on_dev and not capturing and _slots_out is None and img_cipher.shape[0] % self.BATCH_TILE and img_cipher.dtype == torch.float32 and self._has_tiled_conv(keyed):
            # The conv-taps kernels tile the batch in 128 / 256 columns: an odd batch runs their ragged forms (VGG-16, stored order: 186 ms at 64 images, 131 ms at 192, against
            # 72 ms at 128 and 122 ms at 256; matrix cores: 73 ms at 192 against 56 ms at 256 -- profiles/r06_vgg16_other_batches.txt).  Such a batch is padded with zero
            # images to whole tiles, in the feature-major layout the kernels read; every image is its own column of every product, so its logits are what they are in any
            # batch (bit for bit under the stored-order contract), and zero images raise no layer's max |x|.
            n = img_cipher.shape[0]
            padded = -(-n // self.BATCH_TILE) * self.BATCH_TILE
            xp = torch.zeros((img_cipher.shape[1], padded), dtype=torch.float32, device=img_cipher.device).t()
            xp[:n] = img_cipher.detach()
            self._padded_forwards = getattr(self, '_padded_forwards', 0) + 1
            return self.forward_linear(xp, overlap=overlap)[:n]
        y = None
        for _ in range(self.RESCREEN_MAX_PASSES):
            screened = set()
            if on_dev and self.RESCREEN and os.environ.get('KN_NO_RESCREEN') != '1' and not any(getattr(c, '_exact', True) == 'auto' for c in keyed):
                screened = set(k for (k, c) in enumerate(keyed) if c.screened())
            slots = torch.zeros(len(keyed) + 1, dtype=torch.float32, device=img_cipher.device) if screened else None
            y = self._forward_once(img_cipher, overlap, slots, screened)
            if slots is None:
                return y
            if capturing:
                if _slots_out is not None:
                    _slots_out.append((slots, screened))        # a graph replay checks them after the launch (KeyedModel.capture)
                return y
            if os.environ.get('KN_RESCREEN_NOREAD') == '1':      # DIAGNOSTIC (tools/ab_rescreen.py): gather the maxima but skip the host read -- what the read itself costs
                return y
            if not self._rescreen(slots.tolist(), keyed, screened):
                return y
        return y

    def _rescreen(self, xmax, keyed, screened):
        """Host side of the per-forward screen: layers whose input magnitude has outgrown their calibration go back to 'auto' (decided
        again by the next forward, on that batch).  Returns their indices."""
        redo = [k for k in sorted(screened) if keyed[k].rescreen(xmax[k])]
        for k in redo:
            c = keyed[k]
            klayer._log.info('keynet_amd: %s: max |x| = %.3g against %.3g at calibration: re-calibrating on this batch', c._repr, xmax[k], c._contract_record['max_abs_x'])
            c._exact = 'auto'
            c.__dict__.pop('_contract_record', None)
        if redo:
            self.__dict__['_recalibrations'] = self.__dict__.get('_recalibrations', 0) + len(redo)
            self.__dict__.pop('_overlap_plans', None)
        return redo

    BATCH_TILE = 128           # forward_linear pads a device batch of a tiled-conv key-net to whole multiples of this many images
    MAX_BLOCK_ELEMENTS = 1 << 31   # ... and splits a batch whose largest layer would hold this many activations or more into passes

    @staticmethod
    def _has_tiled_conv(keyed):
        return any(isinstance(c.W, ksp.Conv2dTiledMatrix) for c in keyed)

    def _forward_once(self, img_cipher, overlap, slots, screened):
        """One pass over the keyed layers.  `slots` (device f32 [L + 1], zeroed) / `screened` (indices of the keyed layers whose contract
        is re-screened): slot k receives max |x| of keyed layer k -- slot 0 by one pass over the input, slot k + 1 by the kernel that
        produces layer k's output."""
        forced = overlap is True
        if overlap is None and os.environ.get('KN_NO_OVERLAP') == '1':      # A/B switch
            overlap = False
        if any(getattr(c, '_exact', True) == 'auto' for c in self._keynet.children() if isinstance(c, klayer.KeyedLayer)):
            overlap = False        # first forward of an 'auto' key-net: the layers calibrate their contract one by one (KeyedLayer._calibrate)
        if overlap is None:
            overlap = (img_cipher.is_cuda and img_cipher.dtype == torch.float32 and img_cipher.dim() == 2 and img_cipher.shape[0] >= 256 and
                       img_cipher.shape[0] % 256 == 0 and img_cipher.t().is_contiguous() and not torch.cuda.is_current_stream_capturing())
        if not forced and img_cipher.is_cuda and img_cipher.dim() == 2 and not screened:
            chain = self._chain_op(img_cipher.device)
            if chain is not None:
                return self._forward_chain(img_cipher, chain)
        if slots is not None and 0 in screened:
            klayer._absmax_into(img_cipher.detach().t().float(), slots[0:1])
        if overlap:
            plan = self._overlap_plan(img_cipher.device, img_cipher.shape[0], force=forced)
            if plan is not None and img_cipher.is_cuda and img_cipher.dtype == torch.float32 and img_cipher.t().is_contiguous():
                return self._forward_overlapped(img_cipher.detach(), plan, slots, screened)
        children = list(self._keynet.children())
        y = img_cipher
        i = 0
        k = 0
        while i < len(children):
            c = children[i]
            if isinstance(c, klayer.KeyedLayer):
                fuse = (i + 1 < len(children)) and isinstance(children[i + 1], nn.ReLU)
                y = c.forward(y, fuse_relu=fuse, absmax=slots[k + 1:k + 2] if (slots is not None and (k + 1) in screened) else None)
                i += 2 if fuse else 1
                k += 1
            elif isinstance(c, nn.ReLU):
                y = _relu_block(y)
                i += 1
            else:
                raise ValueError('unsupported module in a key-net: %s' % str(type(c)))
        return y

    # -- whole-net kernel: every operator of a small untiled key-net in ONE launch, activations in LDS (csrc/kn_chain.hip) --------
    CHAIN_LDS_BYTES = 160 * 1024
    CHAIN_MAX_OPS = 12

    def _chain_op(self, device):
        """kn_chain_create handle for this key-net on `device`, or None when it does not qualify: every layer a keyed layer under the
        bit-exact contract whose operator is a plain / tiled CSR container, every ReLU fusable into its producer, at most 12 operators,
        and the activations of four batch columns of any two consecutive layers within the CU's 160 KiB of LDS (LeNet_AvgPool: 92 KB).
        KN_NO_CHAIN=1 (A/B switch, read per call) selects the launch-per-layer forward instead."""
        if os.environ.get('KN_NO_CHAIN') == '1':
            return None
        children = list(self._keynet.children())
        sig = tuple((id(c.W), getattr(c, '_exact', True)) if isinstance(c, klayer.KeyedLayer) else None for c in children)
        cache = self.__dict__.setdefault('_chain_ops', {})
        hit = cache.get(device.index)
        if hit is not None and hit[0] == sig:
            return hit[1]
        if torch.cuda.is_current_stream_capturing():
            return None                  # building the chain allocates: not inside a HIP-graph capture (KeyedModel.capture runs an eager forward first)
        op = None
        steps = []
        i = 0
        ok = 0 < len(children)
        while ok and i < len(children):
            c = children[i]
            if (not isinstance(c, klayer.KeyedLayer) or getattr(c, '_exact', True) is not True or isinstance(c.W, ksp.Conv2dTiledMatrix) or not isinstance(c.W, ksp.SparseMatrix) or
                    c.W.is_float64()):                  # (a float64 operator computes in float64: its own row kernel, one launch per layer)
                ok = False
                break
            fuse = (i + 1 < len(children)) and isinstance(children[i + 1], nn.ReLU)
            steps.append((c.W, _capi.KN_FLAG_RELU if (fuse or c.iskeyedrelu()) else 0))
            i += 2 if fuse else 1
        if ok and len(steps) <= self.CHAIN_MAX_OPS:
            feat = [0, 0]
            for (l, (W, _)) in enumerate(steps):
                feat[l & 1] = max(feat[l & 1], int(W.shape[1]))
                feat[(l & 1) ^ 1] = max(feat[(l & 1) ^ 1], int(W.shape[0]))
            if (feat[0] + feat[1] + 1) * 16 <= self.CHAIN_LDS_BYTES and all(steps[l][0].shape[1] == steps[l - 1][0].shape[0] for l in range(1, len(steps))):
                try:
                    with torch.cuda.device(device):
                        op = _capi.Operator.chain([W._device_op(device) for (W, _) in steps], [f for (_, f) in steps])
                except _capi.KeynetHipError as e:
                    # the launch-per-layer forward computes the same thing (bit for bit): a device without 160 KiB of LDS per workgroup, or
                    # no memory left for the packed copy, must not fail the forward.  The failure is cached: no rebuild on every call.
                    klayer._log.warning('keynet_amd: whole-net kernel unavailable for this key-net (%s); using one launch per layer', e)
                    op = None
        cache[device.index] = (sig, op)
        return op

    def _forward_chain(self, x, chain):
        """[N, D0+1] -> [N, classes+1] through the whole-net kernel; stream-ordered on torch's current HIP stream."""
        (rows, cols) = chain.shape()
        assert x.shape[1] == cols, 'Non-conformal shape for the key-net input: %s' % str(tuple(x.shape))
        xt = x.detach().t()
        if xt.dtype != torch.float32:
            xt = xt.float()
        if not xt.is_contiguous():
            xt = xt.contiguous()
        n = xt.shape[1]
        y = torch.empty((rows, n), dtype=torch.float32, device=xt.device)
        with torch.cuda.device(xt.device):
            chain.spmm(xt.data_ptr(), n, n, y.data_ptr(), n, _capi.KN_FLAG_EXACT, torch.cuda.current_stream().cuda_stream)
        return y.t()

    # -- overlapped forward: two half-batch column windows on two HIP streams, one kernel apart ---------------------------------
    OVERLAP_MIN_MACS = 2e11      # below this much work per forward the launches are too short for the overlap to pay (LeNet: launch-bound)
    OVERLAP_PLANS_KEPT = 2       # least-recently-used plans beyond this are dropped (a service that sees many batch sizes must not pile up workspaces)

    def _overlap_plan(self, device, batch, force=False):
        """Launch list + segments for the overlapped forward, or None when this key-net / batch does not qualify.  A layer is run
        per half only if the half batch keeps it on the same kernel instantiation as the whole batch (conv tiles are 128 or 256
        batch columns wide) and it is a matrix-core conv layer; see the segment rule below."""
        key = (device.index, batch, bool(force))
        plans = self.__dict__.setdefault('_overlap_plans', OrderedDict())
        # a plan caches operator handles and flags: it is only valid for the layers' current (operator, contract) identities
        sig = tuple((id(c.W), getattr(c, '_exact', True)) for c in self._keynet.children() if isinstance(c, klayer.KeyedLayer))
        if key in plans and plans[key][0] == sig:
            plans.move_to_end(key)
            return plans[key][1]
        plans.pop(key, None)
        plan = None
        half = batch // 2
        children = list(self._keynet.children())
        big = force or self._macs_per_image() * batch >= self.OVERLAP_MIN_MACS
        f64 = any(isinstance(c, klayer.KeyedLayer) and c.W.is_float64() for c in children)      # float64 operators return float64 blocks: layer by layer
        if big and not f64 and batch % 8 == 0 and half % 128 == 0 and all(isinstance(c, (klayer.KeyedLayer, nn.ReLU)) for c in children):
            steps = []
            i = 0
            while i < len(children) and steps is not None:
                c = children[i]
                if not isinstance(c, klayer.KeyedLayer):
                    steps = None                                   # a ReLU that could not be fused into a producer: simple path
                    break
                fuse = (i + 1 < len(children)) and isinstance(children[i + 1], nn.ReLU)
                contract = getattr(c, '_exact', True)
                if contract == 'split':
                    steps = None                                   # a layer applied in two steps (Conv2dTiledMatrix._split_ops) is not one launch: simple path
                    break
                exact = contract is True or contract == 'auto'
                W = c.W
                relu = fuse or c.iskeyedrelu()
                if type(W) is ksp.SparseMatrix and not exact and W._dense_device_op(device) is not None:
                    (op, ex, ok) = (W._dense_device_op(device), False, half % 128 == 0)
                elif isinstance(W, ksp.Conv2dTiledMatrix):
                    (op, ex) = (W._device_op(device), exact)
                    # order-preserving conv kernels: 256-column tiles (four batch columns per lane).  They also have a 128-column form (two per lane), but
                    # that one is 5-8 % slower per layer (round 5, same-process A/B on every VGG-16 layer shape: conv5_1 4.08 against 3.88 ms) and the
                    # bit-exact forward of 256 images as two overlapped 128-column windows lost 11 % (1 884 against 2 107 images/s): not split.  MFMA tiles
                    # are 128 (Cout > 64) or 256 columns wide
                    ok = (half % 256 == 0) if exact else (half % (128 if W._outshape[0] > 64 else 256) == 0)
                    with torch.cuda.device(device):          # kn_spmm_plan checks the current device like kn_spmm does
                        if contract == 'bf16x3' and 'bf16x3' in op.plan(batch, _capi.KN_FLAG_BF16X3) and 'bf16x3' not in op.plan(half, _capi.KN_FLAG_BF16X3):
                            ok = False     # (cannot happen with 128-column tiles; kept as a guard: a half batch must keep the kernel family)
                else:
                    (op, ex, ok) = (W._device_op(device), True, True)
                flags = (_capi.KN_FLAG_RELU if relu else 0) | (_capi.KN_FLAG_EXACT if ex else 0) | (_capi.KN_FLAG_BF16X3 if (contract == 'bf16x3' and not ex) else 0)
                steps.append((op, int(W.shape[0]), int(W.shape[1]), flags, ok, 'Linear' in c._layertype, isinstance(W, ksp.Conv2dTiledMatrix)))
                i += 2 if fuse else 1
            if steps:
                # ONE split region: from the first layer after which every layer keeps its kernel instantiation on a half batch (VGG at 256
                # images: conv1_1 / conv1_2 need 256-wide tiles and run whole) up to the trailing fully connected layers (too small to fill
                # the chip per half: whole again after the join).  Measured alternatives: joining the streams at every pooling layer so
                # that the pools run whole (a half-batch window halves their gathered row segments: pool1_2 0.93 ms as two halves against
                # 0.79 ms whole) exposes two drains per conv run and is SLOWER than the plain forward (58.96 vs 58.34 ms); nets without
                # matrix-core conv layers have short-lived workgroups and gain nothing (AllConvNet at 4096 images: 108.7 k images/s
                # overlapped vs 109.2 k plain), so for them the plain forward is used.
                join_at = len(steps)
                while join_at > 0 and steps[join_at - 1][5]:
                    join_at -= 1
                split_at = join_at
                while split_at > 0 and steps[split_at - 1][4]:
                    split_at -= 1
                mfma = any(st[6] and not (st[3] & _capi.KN_FLAG_EXACT) for st in steps[split_at:join_at])
                segs = []
                if join_at - split_at >= 2 and (mfma or force):
                    segs = [sg for sg in (('whole', 0, split_at), ('split', split_at, join_at), ('whole', join_at, len(steps))) if sg[2] > sg[1]]
                if any(sg[0] == 'split' for sg in segs):
                    rows_max = max(st[1] for st in steps)
                    plan = dict(steps=steps, segments=segs,
                                bufs=[torch.empty(rows_max * batch, dtype=torch.float32, device=device) for _ in range(2)],
                                streams=[torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)])
        plans[key] = (sig, plan)
        while len(plans) > self.OVERLAP_PLANS_KEPT:      # each plan holds two workspaces of max_rows x batch floats (VGG-16 at 256 images: 2 x 3.3 GB)
            plans.popitem(last=False)
        return plan

    def _macs_per_image(self):
        """Multiply-adds per image of the keyed forward from the host descriptions (factored conv operators count their expansion)."""
        total = 0.0
        for c in self._keynet.children():
            if isinstance(c, klayer.KeyedLayer):
                W = c.W
                if isinstance(W, ksp.Conv2dTiledMatrix) and W._taps is not None:
                    total += float(len(W._taps['ent_out'])) * W._outshape[0] * W._inshape[0]
                else:
                    total += float(W.nnz())
        return total

    def release_workspace(self):
        """Drop the activation workspaces and side streams of the overlapped forward (two ping-pong blocks per batch size)."""
        self.__dict__.pop('_overlap_plans', None)
        self.__dict__.pop('_chain_ops', None)

    def _forward_overlapped(self, x, plan, slots=None, screened=()):
        """x: [N, D0+1] whose transpose is a contiguous feature-major block.  Layers ping-pong between two flat workspaces with the
        SAME leading dimension N, so a stream that owns the column window [c0, c0 + N/2) only ever touches addresses congruent to
        that window modulo N -- the two streams never alias, whatever the layers' row counts.  Stream 1 starts one kernel behind
        stream 0: the two streams' launch boundaries then never coincide, and the workgroups one stream has queued take over the
        CUs that the other stream's draining kernel frees (the drain of a launch costs ~0.3 ms of a 6.5 ms conv layer otherwise).
        Same kernels, same per-column arithmetic: bit-identical to the single-stream forward."""
        (steps, segs, bufs, side) = (plan['steps'], plan['segments'], plan['bufs'], plan['streams'])
        N = x.shape[0]
        half = N // 2
        main = torch.cuda.current_stream(x.device)
        ptr0 = x.t().data_ptr()
        n = len(steps)

        def src_of(k):
            return ptr0 if k == 0 else bufs[(k - 1) % 2].data_ptr()

        def slot_of(k):            # where step k's kernel leaves max |y| (= max |x| of keyed layer k + 1), when that layer is screened
            return (slots.data_ptr() + 4 * (k + 1)) if (slots is not None and (k + 1) in screened) else None

        with torch.cuda.device(x.device):
            if plan.get('done') is not None:
                main.wait_event(plan['done'])                      # the workspaces are shared by successive calls, whatever stream they come from
            for (kind, k0, k1) in segs:
                if kind == 'whole':
                    for k in range(k0, k1):
                        (op, rows, cols, flags) = steps[k][:4]
                        op.spmm(src_of(k), N, N, bufs[k % 2].data_ptr(), N, flags, main.cuda_stream, absmax_ptr=slot_of(k))
                    continue
                for st in side:
                    st.wait_stream(main)
                for k in range(k0, k1 + 1):
                    for (h, st) in enumerate(side):
                        kk = k - h                                 # stream 1 runs one kernel behind stream 0
                        if kk < k0 or kk >= k1:
                            continue
                        (op, rows, cols, flags) = steps[kk][:4]
                        op.spmm(src_of(kk) + 4 * half * h, N, half, bufs[kk % 2].data_ptr() + 4 * half * h, N, flags, st.cuda_stream, absmax_ptr=slot_of(kk))
                    if k == k0:
                        ev = torch.cuda.Event()
                        ev.record(side[0])
                        side[1].wait_event(ev)
                for st in side:
                    main.wait_stream(st)
            rows_out = steps[-1][1]
            out = bufs[(n - 1) % 2][:rows_out * N].view(rows_out, N).clone()      # the workspace is reused by the next call
            plan['done'] = torch.cuda.Event()
            plan['done'].record(main)
        return out.t()

    def exact_mode(self, flag):
        """Switch every keyed layer between the arithmetic contracts WITHOUT re-keying: True = the reference's accumulation
        order and mul-then-add rounding in every layer (bit-exact with scipy: order-preserving kernels, no MFMA); False =
        matrix cores wherever an operator has such a path (conv-taps, large dense operators), whatever the error; 'auto' = per layer,
        decided at the next forward so that the float-key tolerance 1e-5 holds against the reference's arithmetic (KeyedLayer._calibrate;
        see contract_report()); 'auto-bf16x3' = like 'auto', but a conv layer first tries the kernel that emulates f32 products on the bf16
        matrix pipe (three-way split, six of nine cross products: KN_FLAG_BF16X3) and keeps it when its result, measured against the
        order-preserving kernel on the calibration batch, has 4x headroom under the tolerance (EXPERIMENTAL, opt-in); 'bf16x3' forces that
        kernel wherever it applies; None = back to the per-layer setting the key-net was built with (tiled key-nets: 'auto').  Returns self."""
        for c in self._keynet.children():
            if isinstance(c, klayer.KeyedLayer):
                if not hasattr(c, '_exact_built'):
                    c._exact_built = getattr(c, '_exact_decl', getattr(c, '_exact', True))
                if flag == 'auto-bf16x3':        # 'auto' with the bf16x3 kernel as the first candidate (opt-in: never chosen by default)
                    (c._exact, c._allow_bf16x3) = ('auto', True)
                else:
                    c._exact = c._exact_built if flag is None else klayer._contract(flag, True)
                    c._allow_bf16x3 = False
                c.__dict__.pop('_contract_record', None)
        self.__dict__.pop('_overlap_plans', None)                  # the launch lists depend on the layers' contracts
        self.__dict__.pop('_chain_ops', None)
        return self

    def contract_report(self):
        """Per keyed layer: the contract in force (True / False / 'auto' = not decided yet) and, for layers decided by calibration, the
        record of that decision (bound, measured difference, tolerance, max |x| it covers).  `switched` lists the layers calibration moved
        off the matrix cores; `rescreen` says whether every forward re-checks the decisions; `recalibrations` counts the layers a later,
        larger batch sent back to calibration."""
        rows = []
        for (n, c) in self._keynet.named_children():
            if isinstance(c, klayer.KeyedLayer):
                rows.append(dict(name=n, exact=getattr(c, '_exact', True), declared=getattr(c, '_exact_decl', getattr(c, '_exact', True)),
                                 calibration=getattr(c, '_contract_record', None), screened=c.screened()))
        return dict(layers=rows, switched=[r['name'] for r in rows if r['calibration'] is not None and r['calibration'].get('decided') == 'exact' and 'bound' in r['calibration']],
                    undecided=[r['name'] for r in rows if r['exact'] == 'auto'],
                    rescreen=bool(self.RESCREEN and os.environ.get('KN_NO_RESCREEN') != '1' and any(r['screened'] for r in rows)),
                    recalibrations=int(self.__dict__.get('_recalibrations', 0)))

    _LEVEL = {'bf16x3': 0, 'split': 1, False: 2, True: 3}       # codes of a decided contract on the wire (sync_contract); only True (the reference's order) outranks the others

    def sync_contract(self, group=None):
        """COLLECTIVE (every rank of `group` must call it, the same number of times): make the calibration decisions of replicated key-nets
        agree.  Each rank decides a layer's contract on the batches IT sees, so one rank may keep a layer on the matrix cores that another
        rank's larger activations moved to the reference's order; replicas would then no longer be bit-identical.  One all-reduce(MAX) of two
        small integers per keyed layer (the largest and the smallest decided code).  Rule: a layer every deciding rank runs under the SAME
        contract keeps it; any disagreement ends in the reference's order on every rank -- an exact decision anywhere wins, and two different
        re-ordering contracts ('split' on one rank, the fused matrix-core kernel on another) are not ordered by conservativeness: a rank's
        calibration measured ITS kernel on ITS batch only, so neither record covers the other kernel (round-5 advisor finding; until then the
        ranks moved to the larger code and kept the old record).  Returns the names of the layers this rank changed (the caller recomputes its
        current batch when the list is not empty: keynet_amd.dist.sharded_forward does).  Layers still 'auto' (no forward yet) are left alone.
        A no-op without an initialised process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return []
        named = [(n, c) for (n, c) in self._keynet.named_children() if isinstance(c, klayer.KeyedLayer)]
        mine = [self._LEVEL.get(getattr(c, '_exact', True), -1) if getattr(c, '_exact', True) != 'auto' else -1 for (_, c) in named]
        dev = torch.device('cpu') if dist.get_backend(group) == 'gloo' else torch.device('cuda', torch.cuda.current_device())
        t = torch.tensor([mine, [(-m if m >= 0 else -99) for m in mine]], dtype=torch.int32, device=dev)     # row 1: MAX of the negated codes = the smallest decided code
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        (hi, lo) = (t[0].tolist(), [-v for v in t[1].tolist()])
        changed = []
        for ((n, c), m, a, b) in zip(named, mine, hi, lo):
            if m < 0 or a == b or m == self._LEVEL[True]:
                continue                                # undecided here / every deciding rank agrees / already in the reference's order
            rec = dict(getattr(c, '_contract_record', None) or {})
            rec.update(decided='exact', reason=('another rank\'s batch needed the reference\'s order' if a == self._LEVEL[True] else
                                                'ranks calibrated onto different re-ordering kernels: no rank\'s record covers the other\'s') + ' (KeyedModel.sync_contract)')
            rec.pop('max_abs_x', None)                  # nothing left to screen: the reference's order holds for any input
            (c._exact, c._contract_record) = (True, rec)
            changed.append(n)
        if changed:
            self.__dict__.pop('_overlap_plans', None)
            self.__dict__.pop('_chain_ops', None)
        return changed

    def capture(self, img_cipher):
        """Capture forward_linear for this input shape into a HIP graph (torch.cuda.CUDAGraph on ROCm) and return a callable
        `replay(x) -> [N, classes+1]`.  Small key-nets are launch-bound (LeNet at N=1024: 7 kernels in 0.25 ms); one graph
        launch replaces them.  The operators must already be resident and every 'auto' layer decided (one eager forward is run first); the
        returned tensor is the graph's static output buffer (clone it to keep a result across replays).  A key-net with calibrated layers
        keeps its per-forward screen: the graph gathers max |x| per layer like the eager forward, replay() reads it back after the launch
        and, when a layer's input has outgrown its calibration, re-runs the batch eagerly (re-calibrating) and captures a new graph."""
        assert img_cipher.is_cuda, 'capture() needs a device tensor'
        static_in = img_cipher.detach().clone()
        # keep the layout the layers expect: a transposed view of a feature-major block
        if not static_in.t().is_contiguous():
            static_in = static_in.t().contiguous().t()
        state = {}

        def build():
            self.forward_linear(static_in, overlap=False)       # uploads operators, sizes workspaces, calibrates (not capturable)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.forward_linear(static_in, overlap=False)   # warm-up on the capture stream
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            slots = []
            # capture ON THE WARMED STREAM: per-stream state of the operators (the split-K workspace of a dense layer, kn_api.hip) was sized
            # by the warm-up forward above; torch's default capture stream would be a fresh one, and a hipMalloc inside a capture is refused
            with torch.cuda.graph(graph, stream=side):
                out = self.forward_linear(static_in, overlap=False, _slots_out=slots)
            state.update(graph=graph, out=out, slots=slots[0] if slots else None)

        build()

        def replay(x):
            static_in.copy_(x)
            state['graph'].replay()
            if state['slots'] is not None:
                (slots, screened) = state['slots']
                keyed = [c for c in self._keynet.children() if isinstance(c, klayer.KeyedLayer)]
                if self._rescreen(slots.tolist(), keyed, screened):
                    build()                                      # eager forward on this batch re-calibrates; then a fresh graph
                    state['graph'].replay()
            replay.graph = state['graph']
            return state['out']
        replay.graph = state['graph']
        return replay

    def forward(self, img_cipher, outkey=None):
        """Encrypted image(s) [N, D0+1] -> logits.  N == 1 returns the reference's shape `outshape` = (C,1,1)
        (keynet/system.py:130-133); N > 1 (an extension: the reference cannot) returns (N, C, 1, 1)."""
        outkey = outkey if outkey is not None else self.embeddingkey()
        y = self.forward_linear(img_cipher)
        if outkey is not None:
            y = self.decrypt(y, outkey)
        n = y.shape[0]
        return ktorch.linear_to_affine(y, self._outshape if n == 1 else (n,) + tuple(self._outshape))

    def decrypt(self, y_cipher, outkey=None):
        """Apply the embedding key (the reference's version constructs KeyedLayer(W=outkey), an invalid call:
        keynet/system.py:137; the intent -- one more torchdot with the key -- is what is implemented)."""
        outkey = outkey if outkey is not None else self.embeddingkey()
        if outkey is None:
            return y_cipher
        W = outkey if isinstance(outkey, ksp.SparseMatrix) else ksp.SparseMatrix(outkey)
        return W.torchdot(y_cipher.t()).t()

    def imagekey(self):
        return self._imagekey

    def embeddingkey(self):
        return self._embeddingkey

    def public(self):
        """Strip the private keys before releasing the key-net (keynet/system.py:147-151)."""
        (self._imagekey, self._embeddingkey) = (None, None)
        return self

    def num_parameters(self):
        return sum([c.nnz() for (k, c) in self._keynet.named_children() if isinstance(c, klayer.KeyedLayer)])

    def layers(self):
        return self._layernames


def _relu_block(y):
    """Stand-alone nn.ReLU on an [N, D+1] activation (only when it could not be fused into a producer)."""
    if not torch.cuda.is_available():
        raise _capi.KeynetHipError('keynet_amd: no MI355X visible -- the keyed forward has no CPU fallback')
    src = y.device
    yt = y.detach().t().float()
    yt = (yt if yt.is_cuda else yt.cuda()).contiguous().clone()
    _capi.relu(yt.data_ptr(), yt.shape[0], yt.shape[1], yt.shape[1], torch.cuda.current_stream().cuda_stream)
    out = yt.t()
    return out if src.type == 'cuda' else out.to(src)


class KeyedSensor(klayer.KeyedLayer):
    """The paired sensor: applies the image key A_0 to the homogeneous image (keynet/system.py:160-263)."""

    def __init__(self, inshape, keypair):
        assert isinstance(inshape, tuple) and len(inshape) == 3
        nn.Module.__init__(self)
        (self._encryptkey, self._decryptkey) = keypair
        self._inshape = (1, *inshape)
        self._tensor = None
        self._im = None
        self.W = ksp.SparseMatrix(self._encryptkey)
        self._layertype = 'input'
        self._repr = 'KeyedSensor'
        self._tileshape = None
        self._outshape = None

    def __repr__(self):
        return str('<KeyedSensor: height=%d, width=%d, channels=%d>' % (self._inshape[2], self._inshape[3], self._inshape[1]))

    def load(self, imgfile):
        """Read + resize an image file to the sensor shape as a float tensor in [0,255] (keynet/system.py:183-201 via
        vipy; here PIL bilinear -- the resize interpolation is parity-unpinned)."""
        from PIL import Image
        (C, H, W) = self._inshape[1:]
        im = Image.open(imgfile).convert('L' if C == 1 else 'RGB').resize((W, H), Image.BILINEAR)
        a = np.asarray(im, dtype=np.float32)
        a = a.reshape(H, W, 1) if a.ndim == 2 else a
        self._tensor = torch.as_tensor(np.ascontiguousarray(a.transpose(2, 0, 1))).unsqueeze(0)
        return self

    def fromtensor(self, x):
        if x is not None:
            self._tensor = x.clone().float()
        return self

    def tensor(self):
        return self._tensor.unsqueeze(0) if self._tensor.ndim == 3 else self._tensor

    def astensor(self):
        return self.tensor()

    def totensor(self):
        return self.tensor()

    def keypair(self):
        return (self._encryptkey, self._decryptkey)

    def key(self):
        return self._decryptkey

    def isloaded(self):
        return self._tensor is not None

    def isencrypted(self):
        """Encrypted = homogenised [N, C*H*W+1] (the reference only recognises N == 1; batches are an extension)."""
        return self.isloaded() and self._tensor.ndim == 2 and self._tensor.shape[1] == int(np.prod(self._inshape)) + 1

    def encrypt(self):
        """NxCxHxW -> Nx(C*H*W+1) homogenised and keyed (keynet/system.py:250-255)."""
        assert self.isloaded(), 'Load image first'
        if not self.isencrypted():
            self._tensor = self.forward(ktorch.affine_to_linear(self._tensor))
        return self

    def decrypt(self):
        assert self.isloaded(), 'Load image first'
        if self.isencrypted():
            x_raw = super(KeyedSensor, self).decrypt(self._decryptkey, self._tensor)
            n = x_raw.shape[0]
            self._tensor = ktorch.linear_to_affine(x_raw, (n,) + tuple(self._inshape[1:]))
        return self


class PublicKeyedSensor(KeyedSensor):
    """Sensor with the identity key: only homogenises (keynet/system.py:266-284)."""

    def __init__(self, inshape):
        n = int(np.prod(inshape)) + 1
        super(PublicKeyedSensor, self).__init__(inshape, (sparse_identity_matrix(n), sparse_identity_matrix(n)))

    def __repr__(self):
        return str('<PublicKeyedSensor: height=%d, width=%d, channels=%d>' % (self._inshape[2], self._inshape[3], self._inshape[1]))

    def encrypt(self):
        raise ValueError('PublicKeyedSensor has no encryption keys')

    def decrypt(self):
        raise ValueError('PublicKeyedSensor has no decryption keys')

    def tensor(self):
        assert self.isloaded(), 'Load image first'
        if not self.isencrypted():
            self._tensor = self.forward(ktorch.affine_to_linear(self._tensor))
        return self._tensor


# ------------------------------------------------------------------------------------------------------------------
BACKENDS = ('hip',)


def layergen(module, inshape, outshape, A, Ainv, tileshape=None, backend='hip', direct=None, exact=None):
    """The plug-in seam of the reference (keynet/system.py:303-314): snaps the requested tile to divisors of the
    layer's spatial sizes, then dispatches on `backend`.  The reference accepts only 'scipy'; this build registers
    'hip'.  Anything else raises ValueError('invalid backend ...') exactly like the reference."""
    # exact=None: untiled key-nets bit-exact; tiled key-nets 'auto' (matrix cores wherever the 1e-5 float-key contract holds, decided per
    # layer at the first forward: KeyedLayer._calibrate)
    if tileshape is not None:
        tileshape = (find_closest_positive_divisor(outshape[1], tileshape[0]), find_closest_positive_divisor(inshape[1], tileshape[1]))
    if backend == 'hip':
        return klayer.KeyedLayer(module, inshape, outshape, A, Ainv, tileshape=tileshape, direct=direct, exact=exact)
    raise ValueError('invalid backend "%s"' % backend)


def Keynet(inshape, net=None, backend='hip', global_photometric='identity', local_photometric='identity', global_geometric='identity',
           local_geometric='identity', memoryorder='channel', do_output_encryption=False, alpha=None, beta=None, gamma=None,
           hierarchical_blockshape=None, hierarchical_permute_at_level=None, blocksize=None, tileshape=None, direct=None, exact=None):
    """(sensor, model) for `net` under the chosen key family (keynet/system.py:472-486).  ReLU outputs only admit keys
    that commute with ReLU: a 'relu*' layer keeps identity where identity was asked and otherwise falls back to the
    positive-gain / permutation members of the family (keynet/system.py:476-480)."""
    def f_layergen(module, inshape_, outshape_, A, Ainv):
        return layergen(module, inshape_, outshape_, A, Ainv, tileshape=tileshape, backend=backend, direct=direct, exact=exact)

    def f_keypair(layername, shape):
        isrelu = 'relu' in layername
        return keygen(shape,
                      global_photometric=global_photometric if (not isrelu or global_photometric == 'identity') else 'identity',
                      local_photometric=local_photometric if (not isrelu or local_photometric == 'identity') else 'uniform_random_gain',
                      global_geometric=global_geometric if (not isrelu or global_geometric == 'identity') else 'identity',
                      local_geometric=local_geometric if (not isrelu or local_geometric == 'identity') else 'permutation',
                      memoryorder=memoryorder, blocksize=blocksize, tileshape=tileshape, alpha=alpha, beta=beta, gamma=gamma,
                      hierarchical_blockshape=hierarchical_blockshape, hierarchical_permute_at_level=hierarchical_permute_at_level)

    if backend not in BACKENDS:
        raise ValueError('invalid backend "%s"' % backend)
    sensor = KeyedSensor(inshape, f_keypair('input', inshape))
    model = KeyedModel(net, inshape, sensor.key(), f_keypair, f_layergen, do_output_encryption=do_output_encryption) if net is not None else None
    return (sensor, model)


def IdentityKeynet(inshape, net, backend='hip'):
    return Keynet(inshape, net, backend=backend)


def PermutationKeynet(inshape, net, do_output_encryption=False):
    return Keynet(inshape, net, global_geometric='permutation', do_output_encryption=do_output_encryption)


def TiledOrthogonalKeynet(inshape, net, tilesize, hierarchical_permute_at_level=(0, 1), direct=None, exact=None):
    """Hierarchical block permutation + block-local Givens rotations + block-local affine photometric key, block memory
    order (keynet/system.py:504-510): the float-key family (1e-5 contract)."""
    return Keynet(inshape, net, tileshape=(tilesize, tilesize), global_geometric='hierarchical_permutation', hierarchical_blockshape=(2, 2),
                  hierarchical_permute_at_level=hierarchical_permute_at_level, global_photometric='identity', local_geometric='givens_orthogonal',
                  alpha=tilesize, blocksize=tilesize, local_photometric='uniform_random_affine', beta=0.1, gamma=100.0, memoryorder='block',
                  direct=direct, exact=exact)


def TiledIdentityKeynet(inshape, net, tilesize, direct=None, exact=None):
    return Keynet(inshape, net, tileshape=(tilesize, tilesize), direct=direct, exact=exact)


def TiledPermutationKeynet(inshape, net, tilesize, direct=None, exact=None):
    return Keynet(inshape, net, local_geometric='permutation', tileshape=(tilesize, tilesize), blocksize=tilesize, direct=direct, exact=exact)
