"""
Command-line surface of the reference (sparch/parsers/model_config.py:19-65 and
training_config.py:19-147): the same 7 model flags and 19 training flags, names, types, choices
and defaults, so `python run_exp.py ...` invocations carry over unchanged.  Flags ADDED for this build
(no dataset files exist offline; multi-GPU and the bf16 operand mode are new): --synthetic,
--synthetic_batches, --seq_len, --sync_bn, --compute_dtype.  Booleans accept the distutils.strtobool spellings (distutils is gone in Python >= 3.12).
"""
import logging

_TRUE = {"y", "yes", "t", "true", "on", "1"}
_FALSE = {"n", "no", "f", "false", "off", "0"}


def strtobool(value):
    v = str(value).strip().lower()
    if v in _TRUE:
        return True
    if v in _FALSE:
        return False
    raise ValueError(f"invalid truth value {value!r}")


MODEL_FLAGS = [
    # name, type, default, choices, help
    ("model_type", str, "LIF", ["LIF", "adLIF", "RLIF", "RadLIF", "MLP", "RNN", "LiGRU", "GRU"],
     "Neuron / cell type of the network."),
    ("nb_layers", int, 3, None, "Total number of layers, readout included."),
    ("nb_hiddens", int, 128, None, "Width of every hidden layer."),
    ("pdrop", float, 0.1, None, "Dropout probability in [0, 1)."),
    ("normalization", str, "batchnorm", None, "batchnorm, layernorm, or anything else for none."),
    ("use_bias", strtobool, False, None, "Add a trainable bias to the feed-forward projection."),
    ("bidirectional", strtobool, False, None, "Scan the sequence in both directions (doubles layer inputs l>0)."),
]

TRAINING_FLAGS = [
    ("use_pretrained_model", strtobool, False, None, "Load a saved model instead of creating one."),
    ("only_do_testing", strtobool, False, None, "Skip training, only test the loaded model."),
    ("load_exp_folder", str, None, None, "Experiment folder holding the pretrained model (also used for output)."),
    ("new_exp_folder", str, None, None, "Output folder for a new experiment."),
    ("dataset_name", str, "shd", ["shd", "ssc", "hd", "sc"], "Dataset: shd, ssc, hd or sc."),
    ("data_folder", str, "data/shd_dataset/", None, "Dataset location."),
    ("log_tofile", strtobool, False, None, "Write the log to <exp>/log/exp.log instead of the terminal."),
    ("save_best", strtobool, True, None, "Keep the model of the best validation epoch."),
    ("batch_size", int, 128, None, "Examples per batch."),
    ("nb_epochs", int, 5, None, "Number of training epochs."),
    ("start_epoch", int, 0, None, "Epoch to resume from (first trained epoch is start_epoch+1)."),
    ("lr", float, 1e-2, None, "Initial learning rate."),
    ("scheduler_patience", int, 1, None, "Epochs without progress before the learning rate drops."),
    ("scheduler_factor", float, 0.7, None, "Multiplier applied to the learning rate on a plateau."),
    ("use_regularizers", strtobool, False, None, "Penalise firing rates outside [reg_fmin, reg_fmax]."),
    ("reg_factor", float, 0.5, None, "Weight of the firing-rate penalty."),
    ("reg_fmin", float, 0.01, None, "Lowest unpenalised firing rate."),
    ("reg_fmax", float, 0.5, None, "Highest unpenalised firing rate."),
    ("use_augm", strtobool, False, None, "Data augmentation (non-spiking datasets only)."),
]

EXTRA_FLAGS = [
    ("synthetic", strtobool, False, None, "[sparch_amd] use synthetic data of the dataset's shape (no files needed)."),
    ("synthetic_batches", int, 8, None, "[sparch_amd] batches per synthetic epoch."),
    ("seq_len", int, 100, None, "[sparch_amd] time steps of synthetic spiking inputs (loaders bin to 100)."),
    ("sync_bn", strtobool, False, None, "[sparch_amd] data-parallel runs: BatchNorm over the GLOBAL batch "
                                        "(statistics exchanged between ranks) instead of per rank."),
    ("compute_dtype", str, "fp32", ["fp32", "bf16"],
     "[sparch_amd] operand precision of the matrix products: fp32 (exact, the reference's arithmetic) or bf16 "
     "(operands rounded once, fp32 accumulation, states and parameter updates)."),
]


def _add(parser, flags):
    for name, typ, default, choices, text in flags:
        kw = dict(type=typ, default=default, help=text)
        if choices:
            kw["choices"] = choices
        parser.add_argument("--" + name, **kw)
    return parser


def add_model_options(parser):
    return _add(parser, MODEL_FLAGS)


def add_training_options(parser):
    _add(parser, TRAINING_FLAGS)
    return _add(parser, EXTRA_FLAGS)


_MODEL_LABELS = [("Model Type", "model_type"), ("Number of layers", "nb_layers"),
                 ("Number of hidden neurons", "nb_hiddens"), ("Dropout rate", "pdrop"),
                 ("Normalization", "normalization"), ("Use bias", "use_bias"), ("Bidirectional", "bidirectional")]
_TRAIN_LABELS = [("Use pretrained model", "use_pretrained_model"), ("Only do testing", "only_do_testing"),
                 ("Load experiment folder", "load_exp_folder"), ("New experiment folder", "new_exp_folder"),
                 ("Dataset name", "dataset_name"), ("Data folder", "data_folder"), ("Log to file", "log_tofile"),
                 ("Save best model", "save_best"), ("Batch size", "batch_size"), ("Number of epochs", "nb_epochs"),
                 ("Start epoch", "start_epoch"), ("Initial learning rate", "lr"),
                 ("Scheduler patience", "scheduler_patience"), ("Scheduler factor", "scheduler_factor"),
                 ("Use regularizers", "use_regularizers"), ("Regularization factor", "reg_factor"),
                 ("Regularization min firing rate", "reg_fmin"), ("Reguarization max firing rate", "reg_fmax"),
                 ("Use data augmentation", "use_augm")]


def _block(title, labels, args):
    lines = ["", "        " + title, "        " + "-" * len(title)]
    lines += [f"        {label}: {getattr(args, key)}" for label, key in labels]
    return "\n".join(lines) + "\n    "


def print_model_options(args):
    logging.info(_block("Model Config", _MODEL_LABELS, args))


def print_training_options(args):
    logging.info(_block("Training Config", _TRAIN_LABELS, args))
