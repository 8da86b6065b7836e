"""Worker of tests/test_gpu_train.py::test_data_parallel_world2_equals_world1 (one process per rank; several ranks may share
the one GPU of the test box over gloo).  G global episodes are dealt round-robin to the ranks; every rank accumulates the
gradients of its share (weight gradients on the side stream, as in the training loop), then ONE FusedOptimizer.step()
all-reduces the flat bucket and applies SGD.  Rank 0 writes the flat weights; every rank writes its BatchNorm running
statistics; checkpoint_all_ranks() (collective) writes the rank-pooled ones."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, G = sys.argv[1], int(sys.argv[2])
    import litemkd_amd  # noqa: F401
    from litemkd_amd import ops, trainloop as TL
    from litemkd_amd.parallel import init_distributed
    from litemkd_amd.model.model_select import Student, Teacher
    from litemkd_amd.distillers import Distiller
    from litemkd_amd.options import default_args
    from litemkd_amd.utils import aggregate_accuracy
    rank, world, dev = init_distributed()
    if os.environ.get("LMKD_CONV"):      # the arithmetic under test (default: the library's)
        ops.set_conv_compute_dtype(os.environ["LMKD_CONV"])
    cfg = default_args(shot=1, query_per_class=1, img_size=64, trans_dropout=0.0, device=dev, learning_rate=1e-2, save_dir=out_dir,
                       mode="w%d_" % world)
    torch.manual_seed(33)
    student, teacher, _, distiller, aggregate_accuracy, _, opt, _ = TL.make(cfg)      # trainwandb.py:78-109
    w0 = opt.bucket.flat.clone()
    src = TL.SyntheticEpisodes(cfg, base_seed=808, rank=0, device=dev)          # ONE global stream, dealt round-robin
    ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END = True, False
    mine = [e for e in range(G) if e % world == rank]
    for e in mine:
        if e == mine[-1]:
            opt.expect_step()      # world > 1: the tail of the gradient bucket is all-reduced under this backward pass (parallel.EarlyAllReduce)
        TL.train_task(src.episode(e), student, teacher, distiller, aggregate_accuracy, cfg)
    if world > 1 and os.environ.get("LMKD_EXPECT_EARLY", "1") == "1":
        assert opt.early.work is not None and opt.early.split < opt.bucket.numel, "the early all-reduce did not fire"
    opt.step()
    torch.cuda.synchronize()
    bn = {k: v.detach().cpu() for k, v in student.state_dict().items() if "running_" in k}
    torch.save(bn, os.path.join(out_dir, "bn_w%d_r%d.pt" % (world, rank)))
    if rank == 0:
        torch.save({"w0": w0.cpu(), "w1": opt.bucket.flat.detach().cpu()}, os.path.join(out_dir, "flat_w%d.pt" % world))
    TL.checkpoint_all_ranks(student, 1, cfg)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
