"""Learning-rate schedules selectable by ``hparams.lr_schedule`` (reference: lrschedule.py:5-35)."""
import math


def noam_learning_rate_decay(init_lr, global_step, warmup_steps=4000):
    """tensor2tensor "noam" warm-up/decay (lrschedule.py:5-11)."""
    w = float(warmup_steps)
    s = global_step + 1.0
    return init_lr * math.sqrt(w) * min(s * w ** -1.5, s ** -0.5)


def step_learning_rate_decay(init_lr, global_step, anneal_rate=0.98, anneal_interval=100000):
    """lr * rate ** (step // interval) (lrschedule.py:14-17); every shipped preset uses this one."""
    return init_lr * anneal_rate ** (global_step // anneal_interval)


def cyclic_cosine_annealing(init_lr, global_step, T, M):
    """Snapshot-ensemble cyclic cosine (lrschedule.py:20-35)."""
    period = T // M
    return init_lr / 2.0 * (math.cos(math.pi * ((global_step - 1) % period) / period) + 1.0)
