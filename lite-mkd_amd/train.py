"""Training entry point with the reference's command line (trainwandb.py:1-109 `main` / `make`, options.py:7-76 `parse_train_args`,
train_wandb.sh):

    python -m litemkd_amd.train --dataset hmdb --shot 5 --model_backbone resnet18_2fc --model_classifier TRX_2fcsup \
        --model_teacher test_teacher_TRX_2fcsup_fixed --distill_name fc_2_sup_dist --learning_rate 0.0001 \
        --checkpoint_dir hmdb_checkpoint/ --training_iterations 70010 --temp_set 2 --trans_linear_in_dim 2048

Every flag of the reference's parser is accepted under its own name and default (so train_wandb.sh's command line parses unchanged);
the defaults of the three plugin names are train_wandb.sh's values instead of the parser's (`strm18_student` / `TRX` / `test_teacher`
are outside this build's scope).  What the reference does around the loop and this build does not: wandb logging (a
JSON-lines log instead, --log_jsonl), JPEG decoding (episodes come from SyntheticEpisodes or, with --data_dir, from a directory of decoded
uint8 clips through the GPU frame transform).  One process per GPU: under torchrun every rank runs this script on its own episode stream
and the optimizer step all-reduces the gradient bucket (parallel.py).

Extra flags (not in the reference): --dtype, --two_call, --serial (schedule.Schedule), --data_dir / --feature_dir, --log_jsonl,
--seed, --no_save."""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

from . import trainloop as TL
from .options import DEFAULT_CFG
from .schedule import Schedule


def _bool(v):
    """the reference declares `type=bool` flags (options.py:40,52): any non-empty string is True there - `--debug False` included; here
    the usual spellings of false are honoured"""
    return str(v).lower() not in ("", "0", "false", "no", "none")


def build_parser():
    """options.py:7-76: parse_common_args + parse_train_args, flag for flag"""
    p = argparse.ArgumentParser(prog="python -m litemkd_amd.train", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    # few-shot setting
    p.add_argument("--way", type=int, default=5, help="Way of each task.")
    p.add_argument("--shot", type=int, default=5, help="Shots per class.")
    p.add_argument("--query_per_class", type=int, default=5, help="Target samples (i.e. queries) per class used for training.")
    p.add_argument("--query_per_class_test", type=int, default=1, help="Target samples (i.e. queries) per class used for testing.")
    # rarely changed
    p.add_argument("--tasks_per_batch", type=int, default=16, help="Number of tasks between parameter optimizations.")
    p.add_argument("--print_freq", type=int, default=10, help="print and log every n iterations.")
    p.add_argument("--seq_len", type=int, default=8, help="Frames per video.")
    p.add_argument("--num_workers", type=int, default=1, help="Num dataloader workers.")
    p.add_argument("--trans_linear_out_dim", type=int, default=1152, help="Transformer linear_out_dim")
    p.add_argument("--trans_linear_in_dim", type=int, default=2048, help="Transformer linear_in_dim")
    p.add_argument("--img_size", type=int, default=224, help="Input image size to the CNN after cropping.")
    p.add_argument("--temp_set", nargs="+", type=int, default=[2], help="cardinalities e.g. 2,3 is pairs and triples")
    p.add_argument("--trans_dropout", type=float, default=0.1, help="Transformer dropout")
    p.add_argument("--save_freq", type=int, default=10000, help="Number of iterations between checkpoint saves.")
    p.add_argument("--split", type=int, default=3, help="Dataset split.")
    p.add_argument("--sch", nargs="+", type=int, default=[20000, 40000], help="iters to drop learning rate")
    p.add_argument("--num_test_tasks", type=int, default=5000, help="number of random tasks to test on.")
    p.add_argument("--device", default=dev, help="device")
    # changed per experiment
    p.add_argument("--method", choices=["resnet18", "resnet34", "resnet50"], default="resnet18", help="method")
    p.add_argument("--num_gpus", type=int, default=1, help="(reference: GPUs to split the ResNet over with DataParallel; here ranks come from torchrun)")
    p.add_argument("--dataset", choices=["ssv2", "kinetics", "hmdb", "ucf"], default="kinetics", help="Dataset to use.")
    p.add_argument("--mode", default="KD_KL_meta", help="experiment description")
    p.add_argument("--debug", type=_bool, default=False, help="debug mode: no checkpoints")
    p.add_argument("--distill_name", default="fc_2_sup_dist", help="distill experiment name (a Distiller method)")
    p.add_argument("--model_backbone", default="resnet18_2fc", help="backbone name")
    p.add_argument("--model_classifier", default="TRX_2fcsup", help="classifier name")
    p.add_argument("--model_teacher", default="test_teacher_TRX_2fcsup_fixed", help="teacher name")
    p.add_argument("--teacher_checkpoint", default=None, help="teacher checkpoint (MFM layout)")
    p.add_argument("--test_model", choices=["teacher", "student", "extract_feature"], default="student", help="test who")
    p.add_argument("--soft_loss_weight", default=1, help="experiment hyperparameter")
    p.add_argument("--hard_loss_weight", default=1, help="experiment hyperparameter")
    p.add_argument("--test", type=_bool, default=False, help="experiment hyperparameter")
    p.add_argument("--cfg", type=json.loads, default=dict(DEFAULT_CFG), help="loss weights / temperature as a JSON object")
    # parse_train_args
    p.add_argument("--checkpoint_dir", "-c", default=None, help="Directory to save checkpoint to.")
    p.add_argument("--training_iterations", "-i", type=int, default=100010, help="Number of meta-training iterations.")
    p.add_argument("--resume_from_checkpoint", "-r", dest="resume_from_checkpoint", default=False, action="store_true", help="Restart from latest checkpoint.")
    p.add_argument("--test_iters", nargs="+", type=int, default=[10000, 15000, 20000, 30000, 35000, 40000, 50000, 60000, 70000, 80000, 90000, 100000],
                   help="iterations to test at.")
    p.add_argument("--learning_rate", "-lr", type=float, default=0.0001, help="Learning rate.")
    p.add_argument("--opt", choices=["adam", "sgd"], default="sgd", help="Optimizer")
    # ---- not in the reference
    p.add_argument("--dtype", choices=["f32", "f32h2", "f32x3", "f32native", "bf16"], default="f32", help="arithmetic: f32 = f32h2 (default, what bench.py times): "
                   "fp32 tensors and accumulation, the trunk's convolutions on two fp16 planes + power-of-two scales from the tensors' maxima (DESIGN 10): "
                   "full precision for elements within 2^17 of their tensor's maximum, an absolute 2^-40 of the maximum below; a run-time range fence "
                   "(ops.h2_fence_step, once per episode) counts what that leaves under-resolved and moves a tensor whose under-resolved elements would add more "
                   "than a quarter to its rounding error to f32x3's kernels from the next episode on (INTEGRATION.md, 'range fence'); "
                   "f32x3 = every fp32 product from an exact 3-way bf16 split, six products (the library's default arithmetic); "
                   "f32native = v_mfma_f32_32x32x2_f32, bf16 = bf16 tensors + bf16 MFMA (the reference's autocast path, trainwandb.py:20,126)")
    p.add_argument("--no_stream_inputs", dest="stream_inputs", action="store_false", help="with --data_dir: read, upload and transform every episode on "
                   "the training thread (default: a prefetch thread + copy stream + three static input sets, trainloop.StreamedEpisodes; needs one "
                   "frame resolution per episode)")
    p.add_argument("--serial", action="store_true", help="single-stream schedule (Schedule.serial())")
    p.add_argument("--two_call", action="store_true", help="round 3's schedule (two trunk calls on two streams, no cross-episode pipelining) instead of "
                   "the default merged + pipelined one (Schedule.bench())")
    p.add_argument("--data_dir", default=None, help="directory of decoded clips: <data_dir>/<class>/<video>.npy, uint8 [T, H, W, 3]")
    p.add_argument("--feature_dir", default=None, help="teacher features: <feature_dir>/<class>/<video>.npy, float [seq_len, 2048] (default: next to the clips, <video>.feature.npy)")
    p.add_argument("--log_jsonl", default=None, help="append one JSON line per print_freq iterations (the reference logs to wandb)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no_save", action="store_true", help="write no checkpoints")
    return p


def args_check(args):
    """options.py:86-112 `args_cheak` as far as it concerns the hot path"""
    if args.checkpoint_dir is None and not (args.debug or args.no_save):
        print("need to specify a checkpoint dir")
        sys.exit(1)
    if args.model_backbone.startswith("resnet"):          # every backbone of this build projects to 2048 (resnet18_2fc.py:27)
        args.trans_linear_in_dim = 2048
    args.save_dir = args.checkpoint_dir or "model_save"
    args.test_model_path = None
    if args.debug or args.no_save:
        args.save_freq = 0
    return args


class ClipDirectoryEpisodes:
    """Episodes from a directory of DECODED clips (video_reader.VideoDataset.__getitem__, video_reader.py:398-485, with the JPEG decoding
    done beforehand): <root>/<class>/<video>.npy = uint8 [T, H, W, 3]; teacher features [seq_len, 2048] float next to them
    (<video>.feature.npy) or under `feature_root`.  The class / video / frame draws follow the reference (`random.sample` of `way` classes,
    of shot + queries videos per class, get_seq's start / end / linspace frame choice :345-375, the two shuffles :454-460); Resize ->
    crop / flip -> ToTensor run on the GPU (video_transform.GpuFrameTransform), so `support_set` / `target_set` arrive as NHWC4 float
    tensors, which the backbones accept in place of [F, 3, S, S]."""

    def __init__(self, config, root, feature_root=None, device="cuda", length=10 ** 9, train=True):
        from .video_transform import GpuFrameTransform
        self.c, self.root, self.froot, self.device, self.length, self.train = config, root, feature_root, device, length, train
        self.dataset = self
        self.classes = {}
        for cname in sorted(os.listdir(root)):
            d = os.path.join(root, cname)
            if os.path.isdir(d):
                vids = sorted(f[:-4] for f in os.listdir(d) if f.endswith(".npy") and not f.endswith(".feature.npy"))
                if vids:
                    self.classes[cname] = vids
        if len(self.classes) < config.way:
            raise ValueError("%s holds %d classes, the episodes need %d" % (root, len(self.classes), config.way))
        self.tf = GpuFrameTransform(config.img_size, device)

    def _frames(self, clip):
        n, L = clip.shape[0], self.c.seq_len
        if n == L:
            return clip
        if self.train:
            excess_pad = int(min(5, (n - L) / 2))
            if excess_pad < 1:
                start, end = 0, n - 1
            else:
                start = random.randint(0, excess_pad)
                end = random.randint(n - 1 - excess_pad, n - 1)
        else:
            start, end = 1, n - 2
        if end - start < L:
            start, end = 0, n - 1
        idxs = [int(f) for f in np.linspace(start, end, num=L)]
        return clip[idxs]

    def _feature(self, cname, vid):
        path = (os.path.join(self.froot, cname, vid + ".npy") if self.froot else os.path.join(self.root, cname, vid + ".feature.npy"))
        return torch.from_numpy(np.load(path)).float()

    def host_episode(self):
        """the HOST half of episode() for trainloop.StreamedEpisodes (a prefetch thread calls it): the same draws in the same order - classes,
        videos, frames, the two shuffles, then per video flip / crop (support videos first) - but no device work: the decoded frames of all
        videos as ONE uint8 tensor (they must share a resolution: one H2D copy, one resize launch), the crop parameters, the features"""
        c = self.c
        nq = c.query_per_class if self.train else c.query_per_class_test
        batch_classes = random.sample(list(self.classes), c.way)
        sup, tgt = [], []
        for bl, bc in enumerate(batch_classes):
            vids = self.classes[bc]
            if len(vids) < c.shot + nq:
                raise ValueError("class %s has %d videos, an episode needs %d" % (bc, len(vids), c.shot + nq))
            idxs = random.sample(range(len(vids)), c.shot + nq)
            for j, i in enumerate(idxs):
                clip = torch.from_numpy(self._frames(np.load(os.path.join(self.root, bc, vids[i] + ".npy"), mmap_mode="r")).copy())
                (sup if j < c.shot else tgt).append((clip, self._feature(bc, vids[i]), bl))
        random.shuffle(sup)
        random.shuffle(tgt)
        vids = [v[0] for v in sup + tgt]
        if len({tuple(v.shape[1:3]) for v in vids}) != 1:
            raise ValueError("StreamedEpisodes needs one frame resolution per episode (got %s): use the loader without --stream_inputs"
                             % sorted({tuple(v.shape[1:3]) for v in vids}))
        params = [self.tf.draw(v.shape[1], v.shape[2], self.train) for v in vids]
        return {"frames": torch.cat(vids, 0), "params": params, "features": torch.stack([v[1] for v in sup + tgt]), "ns": len(sup),
                "support_labels": torch.FloatTensor([v[2] for v in sup]), "target_labels": torch.FloatTensor([v[2] for v in tgt]),
                "real_target_labels": torch.FloatTensor([v[2] for v in tgt]), "batch_class_list": torch.arange(c.way).float()}

    def episode(self):
        c = self.c
        nq = c.query_per_class if self.train else c.query_per_class_test
        batch_classes = random.sample(list(self.classes), c.way)
        sup, tgt = [], []
        for bl, bc in enumerate(batch_classes):
            vids = self.classes[bc]
            if len(vids) < c.shot + nq:
                raise ValueError("class %s has %d videos, an episode needs %d" % (bc, len(vids), c.shot + nq))
            idxs = random.sample(range(len(vids)), c.shot + nq)
            for j, i in enumerate(idxs):
                clip = torch.from_numpy(self._frames(np.load(os.path.join(self.root, bc, vids[i] + ".npy"), mmap_mode="r")).copy())
                (sup if j < c.shot else tgt).append((clip, self._feature(bc, vids[i]), bl))
        random.shuffle(sup)
        random.shuffle(tgt)
        # one transform call per set: the per-video flip / crop draws come in the reference's order (support videos, then query videos)
        d = {"support_set": self.tf([s[0] for s in sup], train=self.train), "target_set": self.tf([t[0] for t in tgt], train=self.train),
             "support_set_feature_teacher": torch.stack([s[1] for s in sup]).to(self.device), "target_set_feature_teacher": torch.stack([t[1] for t in tgt]).to(self.device),
             "support_labels": torch.FloatTensor([s[2] for s in sup]), "target_labels": torch.FloatTensor([t[2] for t in tgt]),
             "real_target_labels": torch.FloatTensor([t[2] for t in tgt]), "batch_class_list": torch.arange(c.way).float()}
        return {k: v.unsqueeze(0) for k, v in d.items()}

    def __iter__(self):
        for _ in range(self.length):
            yield self.episode()


def schedule_from_args(args):
    conv = {"f32": "fp32h2", "f32h2": "fp32h2", "f32x3": "fp32x3", "f32native": "fp32", "bf16": "bf16"}[args.dtype]
    act = "bf16" if args.dtype == "bf16" else "fp32"
    if args.serial:
        return Schedule.serial(conv_dtype=conv, act_dtype=act)
    if args.two_call or conv == "fp32":          # the native fp32 MFMA mode has no two-segment kernels
        return Schedule.two_call(conv_dtype=conv, act_dtype=act)
    return Schedule.from_env(conv_dtype=conv, act_dtype=act)


def main(argv=None):
    args = args_check(build_parser().parse_args(argv))
    from .parallel import init_distributed
    rank, world, dev = init_distributed()
    args.device = dev
    random.seed(args.seed + rank)
    np.random.seed(args.seed + rank)
    torch.manual_seed(args.seed)                  # identical initial weights on every rank (make() also broadcasts them)
    sched = schedule_from_args(args)
    loader = None
    if args.data_dir:
        loader = ClipDirectoryEpisodes(args, args.data_dir, args.feature_dir, device=dev)
        if args.stream_inputs and torch.device(dev).type == "cuda":
            # the loader beside the compute (trainwandb.py:87-88: a DataLoader worker): a prefetch thread reads and pins the next episodes,
            # a copy stream uploads the uint8 frames and runs the frame transform into one of three static input sets
            loader = TL.StreamedEpisodes(loader, args, dev)
    student, teacher, loader, distiller, accuracy_fn, _, optimizer, scheduler = TL.make(args, video_loader=loader, base_seed=args.seed, schedule=sched)
    if args.teacher_checkpoint:
        from .model.model_select import load_teacher
        load_teacher(teacher.classifier if hasattr(teacher, "classifier") else teacher, args)
    t0 = time.time()

    def log(iteration, a, b):
        rec = ({"iteration": iteration, "loss": a, "accuracy": b} if b is not None else {"iteration": iteration, "test": a})
        rec.update(time=round(time.time() - t0, 3), rank=rank, world=world, lr=optimizer.lr)
        if rank == 0:
            print(json.dumps(rec, default=float), flush=True)
            if args.log_jsonl:
                with open(args.log_jsonl, "a") as f:
                    f.write(json.dumps(rec, default=float) + "\n")
    losses, accs = TL.train(student, teacher, loader, distiller, optimizer, scheduler, accuracy_fn, args, log=log, schedule=sched)
    if not (args.debug or args.no_save):
        path = TL.checkpoint_all_ranks(student, len(losses), args)      # trainwandb.py:183-186: the final checkpoint
        if rank == 0:
            print(json.dumps({"checkpoint": path}), flush=True)
    return losses, accs


if __name__ == "__main__":
    main()
