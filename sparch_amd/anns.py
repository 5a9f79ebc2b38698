"""
Non-spiking baselines with the reference's module API (sparch/models/anns.py).

`ANN`, `MLPLayer`, `RNNLayer`, `LiGRULayer`, `GRULayer`, `ReadoutLayerANN` keep the reference's names,
constructor signatures, parameter names / shapes (state_dict keys) and the order of RNG draws at
construction (anns.py:57-131, 173-208, 255-293, 367-410, 490-538, 617-642).  The arithmetic of MLP layers
and of the readout runs in libsparch_hip.so (projection GEMMs on the exact bf16 split, BatchNorm folded
into the activation kernel, softmax-sum readout) and so does the RNN baseline's cell (the persistent dense
recurrent kernel, csrc/reccell.hip) and the gated baselines' cells (LiGRU, GRU: persistent kernels with 16 hidden
units per workgroup, csrc/gatedcell.hip, for hidden sizes that are multiples of 32 up to 1024).  Other hidden
sizes of the gated baselines run launch-per-step: a host loop over time with the recurrent products on the
library's GEMMs and the gate arithmetic in `sparch_gate_step` (csrc/annstep.hip).  No CPU fallback anywhere.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as Fn
from .snns import _SpikingLayer  # dropout seed helper


def _make_norm(layer, name, normalization, size):
    if normalization == "batchnorm":
        setattr(layer, name, nn.BatchNorm1d(size, momentum=0.05))
        return True
    if normalization == "layernorm":
        setattr(layer, name, nn.LayerNorm(size))
        return True
    return False


_ANN_KINDS = ("MLP", "RNN", "LiGRU", "GRU")


class ANN(nn.Module):
    """Stack of non-spiking baseline layers (reference ANN, anns.py:19-146).

    forward(x: (batch, time, feat) or 4-D (batch, time, feat, channel)) -> (out, None): out is
    (batch, classes) with the readout layer, else (batch, time, feats); the None stands where SNN returns
    firing rates, so that a trainer can treat both alike (anns.py:146)."""

    def __init__(self, input_shape, layer_sizes, ann_type="MLP", dropout=0.0, normalization="batchnorm",
                 use_bias=False, bidirectional=False, use_readout_layer=True):
        super().__init__()
        if ann_type not in _ANN_KINDS:
            raise ValueError(f"Invalid ann type {ann_type}")
        if bidirectional and ann_type == "MLP":
            raise ValueError("MLP cannot be bidirectional.")
        # attributes callers and checkpoints of the reference see
        self.reshape = len(input_shape) > 3
        self.input_size = float(torch.prod(torch.tensor(input_shape[2:])))
        self.batch_size = input_shape[0]
        self.layer_sizes = layer_sizes
        self.num_layers = len(layer_sizes)
        self.num_outputs = layer_sizes[-1]
        self.ann_type = ann_type
        self.dropout = dropout
        self.normalization = normalization
        self.use_bias = use_bias
        self.bidirectional = bidirectional
        self.use_readout_layer = use_readout_layer
        self.is_snn = False
        self.ann = self._init_layers()

    def _init_layers(self):
        layer_cls = _HIDDEN_CLASSES[self.ann_type]
        hidden_sizes = self.layer_sizes[:-1] if self.use_readout_layer else self.layer_sizes
        width = 1 + int(self.bidirectional)
        layers, fan_in = [], self.input_size
        for index, size in enumerate(hidden_sizes):  # creation order = the reference's RNG draw order
            layer = layer_cls(input_size=fan_in, hidden_size=size, batch_size=self.batch_size,
                              dropout=self.dropout, normalization=self.normalization, use_bias=self.use_bias,
                              bidirectional=self.bidirectional)
            layer._layer_index = index  # decorrelates the dropout masks of the layers
            layers.append(layer)
            fan_in = size * width
        if self.use_readout_layer:
            layers.append(ReadoutLayerANN(input_size=fan_in, output_size=self.num_outputs,
                                          normalization=self.normalization, use_bias=self.use_bias))
        return nn.ModuleList(layers)

    def forward(self, x):
        if self.reshape:
            if x.ndim != 4:
                raise NotImplementedError
            x = x.flatten(2)
        for layer in self.ann:
            x = layer(x)
        return x, None


class _ANNLayer(nn.Module):
    _dropout_seed = _SpikingLayer._dropout_seed

    def _norm_args(self, norm_name="norm"):
        is_bn = self.normalization == "batchnorm"
        norm = getattr(self, norm_name) if self.normalize else None
        if is_bn and self.training:
            norm.num_batches_tracked += 1
        return (norm.weight if norm is not None else None, norm.bias if norm is not None else None,
                norm.running_mean if is_bn else None, norm.running_var if is_bn else None)


class _HiddenANNLayer(_ANNLayer):
    """Shared construction of the four hidden-layer types.  GATES lists the projection / recurrent-matrix
    suffixes in the reference's creation order (that order fixes the RNG draws: every W?/V? pair first, then
    the orthogonal initialisation of the V? matrices — anns.py:196, 279-281, 391-396, 514-522)."""
    GATES = ("",)
    RECURRENT = True
    ACT = nn.Sigmoid

    def __init__(self, input_size, hidden_size, batch_size, dropout=0.0, normalization="batchnorm",
                 use_bias=False, bidirectional=False):
        super().__init__()
        self.input_size, self.hidden_size = int(input_size), int(hidden_size)
        self.dropout, self.normalization, self.use_bias = dropout, normalization, use_bias
        if self.RECURRENT:
            self.bidirectional = bidirectional
            self.batch_size = batch_size * (1 + bidirectional)
        else:
            self.batch_size = batch_size
        self.act_fct = self.ACT()
        for g in self.GATES:
            setattr(self, "W" + g, nn.Linear(self.input_size, self.hidden_size, bias=use_bias))
            if self.RECURRENT:
                setattr(self, "V" + g, nn.Linear(self.hidden_size, self.hidden_size, bias=False))
        if self.RECURRENT:
            for g in self.GATES:
                nn.init.orthogonal_(getattr(self, "V" + g).weight)
        self.normalize = False
        for g in self.GATES:
            self.normalize = _make_norm(self, "norm" + g, normalization, self.hidden_size)
        self.drop = nn.Dropout(p=dropout)

    def _cfg(self, x, dirs):
        p_drop = float(self.dropout) if self.training else 0.0
        ln_width = self.hidden_size if (self.normalization == "layernorm" and self.hidden_size % 4) else None
        return {"normalization": self.normalization, "training": self.training, "dirs": dirs, "p_drop": p_drop,
                "seed": self._dropout_seed(x.device) if p_drop > 0 else 0, "ln_width": ln_width}

    # ---- widths that are not multiples of 4 (the kernels own 4 columns per thread; the reference takes any
    # nb_hiddens, anns.py:149-595): the layer runs zero-padded to the next multiple of 4.  A padded unit has zero
    # input weights and zero recurrent weights in both directions, so it feeds nothing; its outputs are sliced away
    # and, the padding being torch.nn.functional.pad, autograd slices the gradients back by itself.  LayerNorm
    # normalises over the width: its kernels take the true width beside the padded one (cfg["ln_width"]) and leave
    # the padding columns out of the statistics (output and input gradient 0 there).
    def _pad_width(self):
        H = self.hidden_size
        return (H + 3) // 4 * 4 - H

    @staticmethod
    def _padded(t, extra, square=False):
        if t is None or extra == 0:
            return t
        pad = (0, extra, 0, extra) if square else ((0, 0) * (t.ndim - 1) + (0, extra))
        return F.pad(t, pad)

    @staticmethod
    def _padded_rows(w, extra):
        return w if extra == 0 else F.pad(w, (0, 0, 0, extra))

    def _finish(self, y, extra, dirs, padded_running):
        """Slice the padded output per direction; copy the padded running statistics back."""
        for (rm, rv), (rm_p, rv_p) in padded_running:
            if rm is not None and rm_p is not rm:
                with torch.no_grad():
                    rm.copy_(rm_p[:rm.numel()])
                    rv.copy_(rv_p[:rv.numel()])
        if extra == 0:
            return y
        B, T = y.shape[0], y.shape[1]
        H = self.hidden_size
        return y.view(B, T, dirs, H + extra)[..., :H].reshape(B, T, dirs * H)


class MLPLayer(_HiddenANNLayer):
    """anns.py:149-227: y = dropout(sigmoid(norm(W x)))."""
    RECURRENT = False

    def forward(self, x):
        Fn._require_device(x, "input")
        if self.batch_size != x.shape[0]:
            self.batch_size = x.shape[0]
        nw, nb, rm, rv = self._norm_args()
        extra = self._pad_width()
        rm_p, rv_p = self._padded(rm, extra), self._padded(rv, extra)
        cfg = dict(self._cfg(x, 1), act="sigmoid", running_mean=rm_p, running_var=rv_p)
        y = Fn.MLPLayerFn.apply(cfg, x, self._padded_rows(self.W.weight, extra), self._padded(self.W.bias, extra),
                                self._padded(nw, extra), self._padded(nb, extra))
        return self._finish(y, extra, 1, [((rm, rv), (rm_p, rv_p))])


class _RecurrentANNLayer(_HiddenANNLayer):
    KIND = None

    def _rows(self, x):
        dirs = 2 if self.bidirectional else 1
        if self.batch_size != x.shape[0] * dirs:
            self.batch_size = x.shape[0] * dirs
        return dirs

    def forward(self, x):
        """LiGRU / GRU (anns.py:412-447, 540-579) on the HIP kernels (functional.GatedLayerFn)."""
        Fn._require_device(x, "input")
        dirs = self._rows(x)
        mats = {"": "c", "z": "z", "r": "r"}
        params, running = [], {}
        extra, back = self._pad_width(), []
        for g in self.GATES:
            W, V = getattr(self, "W" + g), getattr(self, "V" + g)
            nw, nb, rm, rv = self._norm_args("norm" + g)
            rm_p, rv_p = self._padded(rm, extra), self._padded(rv, extra)
            params += [self._padded_rows(W.weight, extra), self._padded(W.bias, extra), self._padded(nw, extra),
                       self._padded(nb, extra), self._padded(V.weight, extra, square=True)]
            running[mats[g]] = (rm_p, rv_p)
            back.append(((rm, rv), (rm_p, rv_p)))
        cfg = dict(self._cfg(x, dirs), kind=self.KIND, running=running)
        return self._finish(Fn.GatedLayerFn.apply(cfg, x, *params), extra, dirs, back)


class RNNLayer(_RecurrentANNLayer):
    """anns.py:230-339: y_t = sigmoid(norm(W x)_t + V y_{t-1}) on the persistent dense recurrent kernel."""
    KIND = "RNN"

    @property
    def uses_persistent_kernel(self):  # what sparch_amd.dp keys its all-reduce policy on
        return not Fn.rec_step_path(self.hidden_size)

    def forward(self, x):
        Fn._require_device(x, "input")
        dirs = self._rows(x)
        nw, nb, rm, rv = self._norm_args()
        extra = self._pad_width()
        rm_p, rv_p = self._padded(rm, extra), self._padded(rv, extra)
        cfg = dict(self._cfg(x, dirs), act="sigmoid", running_mean=rm_p, running_var=rv_p)
        y = Fn.RNNLayerFn.apply(cfg, x, self._padded_rows(self.W.weight, extra), self._padded(self.W.bias, extra),
                                self._padded(nw, extra), self._padded(nb, extra),
                                self._padded(self.V.weight, extra, square=True))
        return self._finish(y, extra, dirs, [((rm, rv), (rm_p, rv_p))])


class LiGRULayer(_RecurrentANNLayer):
    """anns.py:342-462: z = sigmoid(.), c = relu(.), y = z y + (1-z) c — persistent kernels (csrc/gatedcell.hip)
    for hidden sizes that are multiples of 32 up to 1024, launch-per-step otherwise."""
    KIND, GATES, ACT = "LiGRU", ("", "z"), nn.ReLU
    persistent_units_per_workgroup = 16  # a workgroup owns 16 hidden units (sparch_amd.dp sizes the grid with it)

    @property
    def uses_persistent_kernel(self):
        return Fn.ligru_persistent_ok(self.hidden_size)


class GRULayer(_RecurrentANNLayer):
    """anns.py:465-595: z, r = sigmoid(.), c = tanh(W x + V (r y)), y = z y + (1-z) c — persistent kernels with two
    hand-offs per step (csrc/gatedcell.hip) for hidden sizes that are multiples of 32 up to 1024, launch-per-step
    otherwise."""
    KIND, GATES, ACT = "GRU", ("", "z", "r"), nn.Tanh
    persistent_units_per_workgroup = 16

    @property
    def uses_persistent_kernel(self):
        return Fn.gru_persistent_ok(self.hidden_size)


class ReadoutLayerANN(_ANNLayer):
    """anns.py:598-665: norm(W sum_t softmax(x_t)) -> (B, classes)."""

    def __init__(self, input_size, output_size, normalization="batchnorm", use_bias=False):
        super().__init__()
        self.input_size = int(input_size)
        self.output_size = int(output_size)
        self.normalization = normalization
        self.use_bias = use_bias
        self.W = nn.Linear(self.input_size, self.output_size, bias=use_bias)
        self.normalize = _make_norm(self, "norm", normalization, self.output_size)

    def forward(self, x):
        Fn._require_device(x, "input")
        nw, nb, rm, rv = self._norm_args()
        cfg = {"normalization": self.normalization, "training": self.training, "running_mean": rm, "running_var": rv}
        W = self.W.weight
        extra = (-self.input_size) % 4
        if extra:  # the softmax-sum kernels take 4 features per thread: pad with features whose softmax weight is 0
            x = F.pad(x, (0, extra), value=-1e30)
            W = F.pad(W, (0, extra))
        return Fn.ReadoutANNFn.apply(cfg, x, W, self.W.bias, nw, nb)


_HIDDEN_CLASSES = {"MLP": MLPLayer, "RNN": RNNLayer, "LiGRU": LiGRULayer, "GRU": GRULayer}
