"""
`Experiment` — the trainer behind run_exp.py, API-compatible with the reference's
sparch/exp.py (Experiment(args) / .forward(), exp.py:36-147): experiment folders and naming
(149-189), logging (191-212), model construction (291-339), Adam + ReduceLROnPlateau + cross-entropy
(88-100), train / valid / test epoch loops with the same log lines (341-518), best-model
checkpointing as a whole-module pickle (456-463, 126-133).

The training step itself (exp.py:352-382) runs on the MI355X path: `sparch_amd.SNN` on a HIP device.
What this build adds, not replaces:
  * --synthetic 1: batches of the dataset's shape generated on the fly (no SHD/SSC/HD/SC files exist
    offline); without it SHD/SSC go through sparch_amd.dataloaders (h5py event lists binned on the device),
    HD/SC (torchaudio files) are not covered;
  * hd / sc inputs are raw waveforms turned into 40-bin log-mel features ON THE DEVICE by
    `sparch_amd.fbank` (the reference calls torchaudio's kaldi.fbank per clip on the CPU,
    nonspiking_datasets.py:96, 194);
  * data parallelism when launched under torch.distributed.run: per-rank batch shard, per-layer
    gradient all-reduce over RCCL (`sparch_amd.dp.GradAllReducer`); rank 0 logs and checkpoints.
"""
import errno
import logging
import os
import time
from datetime import timedelta

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim.lr_scheduler import ReduceLROnPlateau

from . import dp, optim
from . import functional as Fn
from .functional import check_status, fbank
from .parsers import print_model_options, print_training_options
from .anns import ANN
from .snns import SNN

logger = logging.getLogger(__name__)

_SPIKING_SETS = {"shd": 20, "ssc": 35}
_AUDIO_SETS = {"hd": 20, "sc": 35}


class _SyntheticLoader:
    """Yields (x, xlens, y) like the reference's collate functions (spiking_datasets.py:80-87,
    nonspiking_datasets.py:105-112): spiking sets give (B, T, 700) float spike counts, audio sets give
    (B, 16000) waveforms in [-1, 1] that the trainer converts to (B, 98, 40) log-mel on the device."""

    def __init__(self, kind, batch_size, n_batches, n_classes, seq_len, seed):
        self.kind, self.batch_size, self.n_batches = kind, batch_size, n_batches
        self.n_classes, self.seq_len, self.seed = n_classes, seq_len, seed
        self._pool = None

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        # the batches are generated ONCE (45 M random numbers per headline batch: 0.2 s of host time each, thirty
        # times a 7 ms training step) and served from pinned host memory every epoch — like a dataset that sits in
        # RAM; each epoch still moves every batch over PCIe
        if self._pool is None:
            self._pool = list(self._generate())
        return iter(self._pool)

    def _generate(self):
        g = torch.Generator().manual_seed(self.seed)
        pin = torch.cuda.is_available()
        for _ in range(self.n_batches):
            y = torch.randint(0, self.n_classes, (self.batch_size,), generator=g)
            if self.kind == "spiking":
                # class-dependent input: a band of 700 // n_classes channels fires at 0.25 instead of 0.04,
                # so that a run on synthetic data has something to learn (labels are not independent of x)
                band = 700 // self.n_classes
                ch = torch.arange(700)[None, :]
                hot = (ch >= y[:, None] * band) & (ch < (y[:, None] + 1) * band)
                rate = torch.where(hot, torch.tensor(0.25), torch.tensor(0.04))[:, None, :]
                # one byte per element: binned spike counts are small integers (spiking_datasets.py:71-78); the
                # trainer expands them on the device (functional.input_from_counts) instead of moving 4x the bytes
                # over PCIe every step as the reference does (exp.py:355-356)
                x = (torch.rand(self.batch_size, self.seq_len, 700, generator=g) < rate).to(torch.uint8)
                xlens = torch.full((self.batch_size,), self.seq_len)
            else:
                t = torch.arange(16000) / 16000.0
                tone = torch.sin(2 * np.pi * (200.0 + 40.0 * y[:, None].float()) * t[None])
                x = 0.1 * (torch.rand(self.batch_size, 16000, generator=g) * 2 - 1) + 0.3 * tone
                xlens = torch.full((self.batch_size,), 98)
            if pin:
                x = x.pin_memory()
            yield x, xlens, y


class Experiment:
    """Training / testing of spiking networks on the four speech-command datasets."""

    def __init__(self, args):
        for k, v in vars(args).items():
            setattr(self, k, v)
        self.synthetic = getattr(args, "synthetic", False)
        self.synthetic_batches = getattr(args, "synthetic_batches", 8)
        self.seq_len = getattr(args, "seq_len", 100)
        self.sync_bn = getattr(args, "sync_bn", False)
        self.compute_dtype = getattr(args, "compute_dtype", "fp32")

        self.rank, self.world, self.local_rank = dp.init_from_env()
        self.is_main = self.rank == 0

        self.init_exp_folders()
        self.init_logging()
        print_model_options(args)
        print_training_options(args)

        if not torch.cuda.is_available():
            raise RuntimeError("sparch_amd: no HIP device visible; this build has no CPU training path")
        if os.environ.get("SPARCH_SHARE_GPU", "0") == "1":
            self.local_rank = 0
        self.device = torch.device("cuda", self.local_rank if self.world > 1 else 0)
        torch.cuda.set_device(self.device)
        logging.info(f"\nDevice is set to {self.device}\n")
        Fn.set_compute_dtype(self.compute_dtype)
        if self.compute_dtype != "fp32":
            logging.info("Matrix products run on bf16-rounded operands with fp32 accumulation (--compute_dtype bf16)")

        self.init_dataset()
        self.init_model()

        self.opt = optim.Adam(self.net.parameters(), self.lr)  # exp.py:89, one launch per step (f-2)
        self.scheduler = ReduceLROnPlateau(optimizer=self.opt, mode="max", factor=self.scheduler_factor,
                                           patience=self.scheduler_patience, min_lr=1e-6)
        self.loss_fn = Fn.CrossEntropyLoss()  # exp.py:100 (one launch for the loss and its gradient)
        self.reducer = None
        if self.world > 1:
            self.reducer = dp.GradAllReducer(self.net, rows_per_rank=self.batch_size // self.world)
            logging.info(f"Gradient all-reduce: {self.reducer.bytes_per_step / 1e6:.2f} MB per step, "
                         f"policy '{self.reducer.policy}' (sparch_amd/dp.py)")
            if self.sync_bn:
                Fn.SYNC_BN = {"group": None, "world": self.world}
                logging.info("BatchNorm statistics are exchanged between ranks (--sync_bn)")
        self.status_check_every = 64  # steps between reads of the kernels' status word (a host sync)

    # ---------------------------------------------------------------------------------- driver
    def forward(self):
        if not self.only_do_testing:
            if self.use_pretrained_model:
                logging.info("\n------ Using pretrained model ------\n")
                best_epoch, best_acc = self.valid_one_epoch(self.start_epoch, 0, 0)
            else:
                best_epoch, best_acc = 0, 0
            logging.info("\n------ Begin training ------\n")
            for e in range(best_epoch + 1, best_epoch + self.nb_epochs + 1):
                self.train_one_epoch(e)
                best_epoch, best_acc = self.valid_one_epoch(e, best_epoch, best_acc)
            logging.info(f"\nBest valid acc at epoch {best_epoch}: {best_acc}\n")
            logging.info("\n------ Training finished ------\n")
            if self.save_best:
                path = f"{self.checkpoint_dir}/best_model.pth"
                if os.path.exists(path):
                    self.net = torch.load(path, map_location=self.device, weights_only=False)  # our own file
                logging.info(f"Loading best model, epoch={best_epoch}, valid acc={best_acc}")
            else:
                logging.info("Cannot load best model because save_best option is "
                             "disabled. Model from last epoch is used for testing.")
        if self.dataset_name in ["sc", "ssc"]:
            self.test_one_epoch(self.test_loader)
        else:
            self.test_one_epoch(self.valid_loader)
            logging.info("\nThis dataset uses the same split for validation and testing.\n")

    # ---------------------------------------------------------------------------------- setup
    def init_exp_folders(self):
        if self.use_pretrained_model:
            exp_folder = self.load_exp_folder
            self.load_path = exp_folder + "/checkpoints/best_model.pth"
            if not os.path.exists(self.load_path):
                raise FileNotFoundError(errno.ENOENT, os.strerror(errno.ENOENT), self.load_path)
        elif self.new_exp_folder is not None:
            exp_folder = self.new_exp_folder
        else:
            name = "_".join([
                self.dataset_name, self.model_type,
                f"{self.nb_layers}lay{self.nb_hiddens}", f"drop{self.pdrop}", str(self.normalization),
                "bias" if self.use_bias else "nobias", "bdir" if self.bidirectional else "udir",
                "reg" if self.use_regularizers else "noreg", f"lr{self.lr}"])
            exp_folder = "exp/test_exps/" + name.replace(".", "_")
        # decided on EVERY rank (the ranks of one node see the same file system) before any collective: a
        # rank-0-only exception would leave the others waiting in the parameter broadcast
        exists = (not self.use_pretrained_model) and os.path.exists(exp_folder)
        if self.world > 1:
            flag = torch.tensor([int(exists)])
            if torch.distributed.get_backend() != "gloo":
                flag = flag.cuda()
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            exists = bool(int(flag.item()))
        if exists:
            raise FileExistsError(errno.EEXIST, os.strerror(errno.EEXIST), exp_folder)
        if self.world > 1:  # nobody creates the folder while a peer is still looking for it
            torch.distributed.barrier()
        self.log_dir = exp_folder + "/log/"
        self.checkpoint_dir = exp_folder + "/checkpoints/"
        if self.is_main:
            os.makedirs(self.log_dir, exist_ok=True)
            os.makedirs(self.checkpoint_dir, exist_ok=True)
        self.exp_folder = exp_folder

    def init_logging(self):
        level = logging.INFO if self.is_main else logging.WARNING
        if self.log_tofile and self.is_main:
            logging.basicConfig(filename=self.log_dir + "exp.log", level=level, format="%(message)s")
        else:
            logging.basicConfig(level=level, format="%(message)s")

    def init_dataset(self):
        if self.dataset_name in _SPIKING_SETS:
            self.nb_inputs, self.nb_outputs, kind = 700, _SPIKING_SETS[self.dataset_name], "spiking"
            if self.use_augm:
                logging.warning("\nWarning: Data augmentation not implemented for SHD and SSC.\n")
        elif self.dataset_name in _AUDIO_SETS:
            self.nb_inputs, self.nb_outputs, kind = 40, _AUDIO_SETS[self.dataset_name], "audio"
        else:
            raise ValueError(f"Invalid dataset name {self.dataset_name}")
        self.input_kind = kind
        if self.batch_size % self.world != 0:
            raise ValueError(f"batch_size {self.batch_size} must be divisible by the number of GPUs {self.world}")
        per_rank = self.batch_size // self.world
        if not self.synthetic:
            if kind != "spiking":
                raise RuntimeError(
                    "sparch_amd: the file-based HD/SC (torchaudio) loader of the reference is not part of this "
                    "build (no torchaudio offline); run with --synthetic 1")
            from .dataloaders.spiking_datasets import load_shd_or_ssc  # exp.py:224-252 (needs h5py + the files)

            def ld(split, shuffle):  # data-parallel: each rank draws its 1/world share of every epoch
                return load_shd_or_ssc(self.dataset_name, self.data_folder, split, per_rank, nb_steps=100,
                                       shuffle=shuffle, device=self.device, rank=self.rank, world=self.world)

            self.train_loader, self.valid_loader = ld("train", True), ld("valid", False)
            if self.dataset_name == "ssc":
                self.test_loader = ld("test", False)
            return

        def mk(seed):
            return _SyntheticLoader(kind, per_rank, self.synthetic_batches, self.nb_outputs, self.seq_len,
                                    seed * 1000 + self.rank)

        self.train_loader, self.valid_loader = mk(1), mk(2)
        if self.dataset_name in ["sc", "ssc"]:
            self.test_loader = mk(3)

    def init_model(self):
        input_shape = (self.batch_size // self.world, None, self.nb_inputs)
        layer_sizes = [self.nb_hiddens] * (self.nb_layers - 1) + [self.nb_outputs]
        if self.use_pretrained_model:
            self.net = torch.load(self.load_path, map_location=self.device, weights_only=False)  # our own file
            logging.info(f"\nLoaded model at: {self.load_path}\n {self.net}\n")
        elif self.model_type in ["LIF", "adLIF", "RLIF", "RadLIF"]:
            self.net = SNN(input_shape=input_shape, layer_sizes=layer_sizes, neuron_type=self.model_type,
                           dropout=self.pdrop, normalization=self.normalization, use_bias=self.use_bias,
                           bidirectional=self.bidirectional, use_readout_layer=True).to(self.device)
            logging.info(f"\nCreated new spiking model:\n {self.net}\n")
        elif self.model_type in ["MLP", "RNN", "LiGRU", "GRU"]:
            # exp.py:311-322: MLP and RNN on the fused / persistent kernels, LiGRU and GRU
            # launch-per-step
            self.net = ANN(input_shape=input_shape, layer_sizes=layer_sizes, ann_type=self.model_type,
                           dropout=self.pdrop, normalization=self.normalization, use_bias=self.use_bias,
                           bidirectional=self.bidirectional, use_readout_layer=True).to(self.device)
            logging.info(f"\nCreated new non-spiking model:\n {self.net}\n")
        else:
            raise ValueError(f"Invalid model type {self.model_type}")
        if self.world > 1:  # identical initial replicas
            for p in self.net.parameters():
                torch.distributed.broadcast(p.data, src=0)
        self.nb_params = sum(p.numel() for p in self.net.parameters() if p.requires_grad)
        logging.info(f"Total number of trainable parameters is {self.nb_params}")

    # ---------------------------------------------------------------------------------- epochs
    def _to_device(self, x, y):
        x = x.to(self.device, non_blocking=True)
        y = y.to(self.device, non_blocking=True)
        if x.dtype == torch.uint8:  # spike counts as bytes: expanded on the device
            x = Fn.input_from_counts(x) if self.net.is_snn else x.float()
        if self.input_kind == "audio":
            x = fbank(x, num_mel_bins=40)  # (B, 16000) -> (B, 98, 40) on the device
        return x, y

    def _prefetched(self, loader):
        """The loader's batches, on the device, uploaded ONE BATCH AHEAD on a side stream (pinned source,
        non-blocking copy) while the compute stream runs the current step; the compute stream waits for a batch's
        event before it uses it.  In stream order (rounds 1-2) the upload sat between two steps."""
        main = torch.cuda.current_stream(self.device)
        side = getattr(self, "_upload_stream", None)
        if side is None:
            side = self._upload_stream = torch.cuda.Stream(device=self.device)

        def put(batch):
            x, xlens, y = batch
            with torch.cuda.stream(side):
                xd = x.to(self.device, non_blocking=True)
                yd = y.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            return xd, xlens, yd, ev

        it = iter(loader)
        try:
            nxt = put(next(it))
        except StopIteration:
            return
        while nxt is not None:
            xd, xlens, yd, ev = nxt
            try:
                nxt = put(next(it))
            except StopIteration:
                nxt = None
            main.wait_event(ev)
            for t in (xd, yd):
                if t.is_cuda:
                    t.record_stream(main)
            yield xd, xlens, yd

    def _mean_over_ranks(self, value):
        if self.world == 1:
            return value
        t = torch.tensor([float(value)], device=self.device, dtype=torch.float64)
        torch.distributed.all_reduce(t)
        return float(t.item()) / self.world

    def train_one_epoch(self, e):
        start = time.time()
        self.net.train()
        if hasattr(getattr(self.train_loader, "sampler", None), "set_epoch"):
            self.train_loader.sampler.set_epoch(e)  # per-rank shards of a fresh permutation every epoch
        losses, accs, sizes = [], [], []
        epoch_spike_rate = 0
        seen = 0
        # No host round trip inside the loop: the reference's per-step `.item()` / `.cpu()`
        # (exp.py:363, 381) become device-side lists read back ONCE per epoch — the same fp32 per-batch
        # values, the same float64 means — and the kernels' status word is checked at the same point.
        for step, (x, _, y) in enumerate(self._prefetched(self.train_loader)):
            x, y = self._to_device(x, y)
            output, firing_rates = self.net(x)
            loss_val = self.loss_fn(output, y)
            losses.append(loss_val.detach())
            if self.net.is_snn:
                epoch_spike_rate += torch.mean(firing_rates)
                if self.use_regularizers:
                    reg_quiet = F.relu(self.reg_fmin - firing_rates).sum()
                    reg_burst = F.relu(firing_rates - self.reg_fmax).sum()
                    loss_val = loss_val + self.reg_factor * (reg_quiet + reg_burst)
            self.opt.zero_grad()
            loss_val.backward()
            if self.reducer is not None:
                self.reducer.finish()
            self.opt.step()
            pred = torch.argmax(output, dim=1)
            accs.append((y == pred).sum())
            sizes.append(y.shape[0])
            seen += x.shape[0] * x.shape[1]
            if (step + 1) % self.status_check_every == 0:
                self._check_kernels(e, step)
        self._check_kernels(e, step)
        losses = torch.stack(losses).cpu().numpy().astype(np.float64) if losses else np.zeros(0)
        accs = (torch.stack(accs).cpu().numpy().astype(np.float64) / np.asarray(sizes, np.float64)) if accs else np.zeros(0)
        logging.info(f"Epoch {e}: lr={self.opt.param_groups[-1]['lr']}")
        logging.info(f"Epoch {e}: train loss={self._mean_over_ranks(np.mean(losses))}")
        logging.info(f"Epoch {e}: train acc={self._mean_over_ranks(np.mean(accs))}")
        if self.net.is_snn:
            epoch_spike_rate /= step  # sic: the reference divides by the last batch index (exp.py:398)
            logging.info(f"Epoch {e}: train mean act rate={epoch_spike_rate}")
        elapsed = time.time() - start
        logging.info(f"Epoch {e}: train elapsed time={str(timedelta(seconds=elapsed))}")
        logging.info(f"Epoch {e}: train throughput={self.world * seen / elapsed:.0f} timesteps*samples/s")

    def _check_kernels(self, e, step):
        """Read the recurrent kernels' status word.  After an in-kernel timeout (the persistent grid could
        not become co-resident: GPU shared or partitioned) the steps since the last check were no-ops on the
        device (the optimizer and the BatchNorm statistics skip themselves while the word is raised); this
        process now continues with one launch per time step — new launches, same process."""
        if self._timeout_on_any_rank():
            logging.warning(f"Epoch {e}, step {step}: {Fn._TIMEOUT_TEXT}  [{Fn.describe_timeout()}]  Steps since the "
                            "timeout were skipped on every rank; continuing with one kernel launch per time step "
                            "(SPARCH_REC_STEPS_PER_LAUNCH=1 behaviour).")

    def _timeout_on_any_rank(self):
        """Collective read of the status word: every rank sees the MAX over ranks (the reducer already merges it
        every training step; evaluation has no reducer), so all ranks degrade together — a rank that raised
        alone used to leave its peers waiting in the next collective.  After a degrade the replicas are
        re-synchronised from rank 0 (they skipped the same steps, so this is a safety net, not a repair)."""
        if self.world > 1:
            dp.sync_status(self.device)
        if not check_status(self.device, on_timeout="degrade"):
            return False
        if self.world > 1:
            for t in list(self.net.parameters()) + list(self.net.buffers()):
                torch.distributed.broadcast(t.data, src=0)
        return True

    def _eval_epoch(self, loader, retried=False):
        losses, accs, sizes = [], [], []
        epoch_spike_rate = 0
        step = 0
        for step, (x, _, y) in enumerate(self._prefetched(loader)):
            x, y = self._to_device(x, y)
            output, firing_rates = self.net(x)
            losses.append(self.loss_fn(output, y).detach())
            pred = torch.argmax(output, dim=1)
            accs.append((y == pred).sum())
            sizes.append(y.shape[0])
            if self.net.is_snn:
                epoch_spike_rate += torch.mean(firing_rates)
        if self._timeout_on_any_rank():  # the epoch's numbers are invalid: once more, now with per-step launches
            if retried:
                raise Fn._capi.SparchHipError(Fn._TIMEOUT_TEXT + "  [" + Fn.describe_timeout() + "]")
            logging.warning(f"Evaluation: {Fn._TIMEOUT_TEXT}  [{Fn.describe_timeout()}]  Repeating the epoch with one "
                            "kernel launch per time step.")
            return self._eval_epoch(loader, retried=True)
        losses = torch.stack(losses).cpu().numpy().astype(np.float64) if losses else np.zeros(0)
        accs = (torch.stack(accs).cpu().numpy().astype(np.float64) / np.asarray(sizes, np.float64)) if accs else np.zeros(0)
        if self.net.is_snn:
            epoch_spike_rate /= step  # sic (exp.py:449, 515)
        return (self._mean_over_ranks(np.mean(losses)), self._mean_over_ranks(np.mean(accs)), epoch_spike_rate)

    def valid_one_epoch(self, e, best_epoch, best_acc):
        with torch.no_grad():
            self.net.eval()
            valid_loss, valid_acc, rate = self._eval_epoch(self.valid_loader)
            logging.info(f"Epoch {e}: valid loss={valid_loss}")
            logging.info(f"Epoch {e}: valid acc={valid_acc}")
            if self.net.is_snn:
                logging.info(f"Epoch {e}: valid mean act rate={rate}")
            self.scheduler.step(valid_acc)
            if valid_acc > best_acc:
                best_acc, best_epoch = valid_acc, e
                if self.save_best and self.is_main:
                    torch.save(self.net, f"{self.checkpoint_dir}/best_model.pth")
                    logging.info(f"\nBest model saved with valid acc={valid_acc}")
            if self.world > 1:
                torch.distributed.barrier()
            logging.info("\n-----------------------------\n")
            return best_epoch, best_acc

    def test_one_epoch(self, test_loader):
        with torch.no_grad():
            self.net.eval()
            logging.info("\n------ Begin Testing ------\n")
            test_loss, test_acc, rate = self._eval_epoch(test_loader)
            logging.info(f"Test loss={test_loss}")
            logging.info(f"Test acc={test_acc}")
            if self.net.is_snn:
                logging.info(f"Test mean act rate={rate}")
            logging.info("\n-----------------------------\n")
