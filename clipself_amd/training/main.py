"""`python -m clipself_amd.training.main ...` -- orchestration with the reference's flow (src/training/main.py:55-342):
parse flags -> init distributed -> student (+ teacher from the same checkpoint) -> lock_image_tower -> data-parallel
wrap -> AdamW param groups -> [resume] -> data -> LR schedule -> epochs { train_one_epoch; alpha weight-space ensemble
with the teacher; checkpoint {epoch,name,state_dict,optimizer}; eval }."""
import logging
import os
import random
import sys
from datetime import datetime

import numpy as np
import torch

from ..open_clip import create_model, create_model_and_transforms
from .clipself import CLIPSelf
from .data import get_data
from .distributed import (FrozenDataParallel, StudentDataParallel, broadcast_object, init_distributed_device, is_master)
from .logger import setup_logging
from .optim import FlatAdamW
from .params import parse_args
from .scheduler import const_lr, const_lr_cooldown, cosine_lr
from .train import evaluate, student_teacher_ensemble, train_one_epoch

LATEST_CHECKPOINT_NAME = "epoch_latest.pt"


def random_seed(seed=42, rank=0):
    torch.manual_seed(seed + rank)
    np.random.seed(seed + rank)
    random.seed(seed + rank)


def build_optimizer(model, args):
    """main.py:198-213"""
    world = args.world_size if args.distributed else 1
    return FlatAdamW(model.module if hasattr(model, "module") else model, lr=args.lr, betas=(args.beta1, args.beta2),
                     eps=args.eps, weight_decay=args.wd, grad_divisor=float(world))


def build_scheduler(optimizer, args, num_batches):
    total_steps = (num_batches // args.accum_freq) * args.epochs
    if args.lr_scheduler == "cosine":
        return cosine_lr(optimizer, args.lr, args.warmup, total_steps)
    if args.lr_scheduler == "const":
        return const_lr(optimizer, args.lr, args.warmup, total_steps)
    if args.lr_scheduler == "const-cooldown":
        assert getattr(args, "epochs_cooldown", None) is not None, "Please specify the number of cooldown epochs for this lr schedule."
        cooldown = (num_batches // args.accum_freq) * args.epochs_cooldown
        return const_lr_cooldown(optimizer, args.lr, args.warmup, total_steps, cooldown, args.lr_cooldown_power, args.lr_cooldown_end)
    raise ValueError(f"Unknown scheduler, {args.lr_scheduler}. Available options are: cosine, const, const-cooldown.")


def _calibrate_frozen(m):
    """Frozen-schedule fold guard of a tower whose weights were just loaded (EvaEngine.calibrate_block_folds: collective in a process group)."""
    eng = getattr(getattr(m, "visual", None), "engine", None)
    if eng is not None and getattr(eng, "block_fold_guard", False) and getattr(eng, "fold_block_ln", False):
        eng.calibrate_block_folds()


def main(argv):
    args = parse_args(argv)
    device = init_distributed_device(args)
    if args.name is None:
        date_str = datetime.now().strftime("%Y_%m_%d-%H_%M_%S")
        if args.distributed:
            date_str = broadcast_object(args, date_str)
        args.name = "-".join([date_str, f"model_{args.model.replace('/', '-')}", f"lr_{args.lr}", f"b_{args.batch_size}",
                              f"j_{args.workers}", f"p_{args.precision}"])
    log_base_path = os.path.join(args.logs, args.name)
    args.log_path = None
    if is_master(args, local=args.log_local):
        os.makedirs(log_base_path, exist_ok=True)
        args.log_path = os.path.join(log_base_path, f"out-{args.rank}" if args.log_local else "out.log")
        if os.path.exists(args.log_path):
            print("Error. Experiment already exists. Use --name {} to specify a new experiment.")
            return -1
    setup_logging(args.log_path, logging.DEBUG if args.debug else logging.INFO)
    args.checkpoint_path = os.path.join(log_base_path, "checkpoints")

    random_seed(args.seed, 0)
    model, preprocess_train, preprocess_val = create_model_and_transforms(
        args.model, args.pretrained, precision=args.precision, device=device, cache_dir=args.cache_dir,
        det_image_size=args.det_image_size, dataset_type=args.dataset_type)
    model.visual.teacher_chunk = args.teacher_chunk
    args.teacher_prefetch = not args.no_teacher_prefetch
    args.input_size = model.visual.image_size
    args.tower_cfg = model.visual.cfg
    if args.dataset_type in ("grid_distill", "proposals_distill"):
        method = CLIPSelf()
        dist_model = create_model(args.model, args.pretrained, device=device, precision=args.precision,
                                  cache_dir=args.cache_dir, trainable=False)
        dist_model.visual.teacher_chunk = args.teacher_chunk
    elif args.dataset_type == "region_clip":
        from .region_clip import RegionCLIP
        nouns = None
        if not (args.train_embed_path and os.path.exists(args.train_embed_path)):
            # the reference's noun-embedding files are not shipped (.MISSING_LARGE_BLOBS): seeded unit vectors stand in
            g = torch.Generator().manual_seed(4764)
            nouns = torch.randn(4764, model.embed_dim, generator=g)
            logging.info("region_clip: --train-embed-path not found, using a seeded synthetic noun bank [4764, E]")
        method = RegionCLIP(args, noun_embeddings=nouns).to(device)
        dist_model = None                                       # main.py:145-147: no teacher for RegionCLIP
    else:
        raise NotImplementedError(args.dataset_type)
    random_seed(args.seed, args.rank)
    if args.lock_image:
        model.lock_image_tower(unlocked_groups=args.lock_image_unlocked_groups, freeze_bn_stats=args.lock_image_freeze_bn_stats)
    elif args.train_data:
        # Without --lock-image the reference trains the whole visual tower -- stem, positional embedding, final norm and head besides the
        # blocks -- through the dense path (main.py:161-166; eva_vit_model.py:537-544,615-623).  A freshly built EVA02 tower is in that
        # state (EvaEngine.set_trainable_all), and so is an OpenAI-CLIP ViT (ClipVitEngine: stem, positional embedding, ln_post and proj
        # train as well; UNLOCKED_TRAINS_ALL on both tower classes).  A tower class without that flag raises.
        if not getattr(model.visual, "UNLOCKED_TRAINS_ALL", False):
            raise NotImplementedError("training without --lock-image is supported for the EVA02 and OpenAI-CLIP ViT towers; pass --lock-image "
                                      "--lock-image-unlocked-groups N for this model family")
        model.visual.unlock()
    if is_master(args):
        with open(os.path.join(args.logs, args.name, "params.txt"), "w") as f:
            for name in sorted(vars(args)):
                logging.info(f"  {name}: {getattr(args, name)}")
                f.write(f"{name}: {getattr(args, name)}\n")
    if args.distributed:
        model = StudentDataParallel(model)
        dist_model = FrozenDataParallel(dist_model) if dist_model is not None else None

    optimizer = build_optimizer(model, args) if args.train_data else None
    start_epoch = 0
    if args.resume is not None:
        checkpoint = torch.load(args.resume, map_location="cpu", weights_only=False)
        target = model.module if hasattr(model, "module") else model
        if "epoch" in checkpoint:
            start_epoch = checkpoint["epoch"]
            sd = checkpoint["state_dict"]
            if next(iter(sd.items()))[0].startswith("module"):
                sd = {k[len("module."):]: v for k, v in sd.items()}
            target.load_state_dict(sd)
            if optimizer is not None:
                optimizer.load_state_dict(checkpoint["optimizer"])
            logging.info(f"=> resuming checkpoint '{args.resume}' (epoch {start_epoch})")
        else:
            target.load_state_dict(checkpoint)

    data = get_data(args, (preprocess_train, preprocess_val), epoch=start_epoch)
    scheduler = build_scheduler(optimizer, args, data["train"].dataloader.num_batches) if optimizer is not None else None
    args.save_logs = args.logs and args.logs.lower() != "none" and is_master(args)
    os.makedirs(args.checkpoint_path, exist_ok=True)
    evaluate(model, data, start_epoch, args)
    if "train" not in data:                                     # main.py:258-263: evaluation-only run (scripts/test_*.sh: --train-data "")
        return 0

    for epoch in range(start_epoch, args.epochs):
        if is_master(args):
            logging.info(f"Start epoch {epoch}")
        train_one_epoch(model, method, data, None, epoch, optimizer, None, scheduler, dist_model, args)
        completed_epoch = epoch + 1
        student_sd = (model.module if args.distributed else model).state_dict()
        if args.alpha < 1.0:
            if dist_model is None:                              # main.py:285-293: re-create the pretrained model for the ensemble
                ref_model = create_model(args.model, args.pretrained, device=device, precision=args.precision,
                                         cache_dir=args.cache_dir, trainable=False)
                teacher_sd = ref_model.state_dict()
                del ref_model
            else:
                teacher_sd = (dist_model.module if args.distributed else dist_model).state_dict()
            target_sd = student_teacher_ensemble(student_sd, teacher_sd, args.alpha)
        else:
            target_sd = student_sd
        eval_due = "val" in data and args.zeroshot_frequency != 0 and (
            completed_epoch % args.zeroshot_frequency == 0 or completed_epoch == args.epochs)       # the condition of zero_shot_eval
        if eval_due:                                            # main.py:300-342: the saved (ensembled) weights are what gets evaluated
            test_model = create_model(args.model, args.pretrained, device=device, precision=args.precision, cache_dir=None,
                                      trainable=False)
            test_model.load_state_dict(target_sd)
            _calibrate_frozen(test_model)                       # every rank is here: the fold guard's collective belongs here, not inside evaluate()
            evaluate(test_model, data, completed_epoch, args)
            del test_model
        if is_master(args):
            ckpt = {"epoch": completed_epoch, "name": args.name, "state_dict": target_sd, "optimizer": optimizer.state_dict()}
            if completed_epoch == args.epochs or (args.save_frequency > 0 and completed_epoch % args.save_frequency == 0):
                torch.save(ckpt, os.path.join(args.checkpoint_path, f"epoch_{completed_epoch}.pt"))
            if args.delete_previous_checkpoint:
                prev = os.path.join(args.checkpoint_path, f"epoch_{completed_epoch - 1}.pt")
                if os.path.exists(prev):
                    os.remove(prev)
            if args.save_most_recent:
                tmp = os.path.join(args.checkpoint_path, "tmp.pt")
                torch.save(ckpt, tmp)
                os.replace(tmp, os.path.join(args.checkpoint_path, LATEST_CHECKPOINT_NAME))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
