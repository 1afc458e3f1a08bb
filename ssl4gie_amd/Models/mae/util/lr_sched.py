"""Per-iteration LR schedule of the MAE drivers (reference: Models/mae/util/lr_sched.py:9-21):
linear warm-up over `warmup_epochs`, then half-cycle cosine to `min_lr`; groups carrying an
`lr_scale` (layer decay) get the scaled value."""
import math


def adjust_learning_rate(optimizer, epoch, args):
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        t = (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * (1.0 + math.cos(math.pi * t))
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr
