"""optimizers/polydecay.py of the reference (:44-76): WarmUpPolyDecay learning-rate schedule."""


class WarmUpPolyDecay:
    def __init__(self, initial_learning_rate, decay_steps, end_learning_rate=0.0001, warmup_steps=0, warmup_learning_rate=1e-4,
                 power=1.0, name=None):
        self.initial_learning_rate = initial_learning_rate
        self.decay_steps = decay_steps
        self.end_learning_rate = end_learning_rate
        self.power = power
        self.warmup_steps = warmup_steps
        self.warmup_learning_rate = warmup_learning_rate
        self.name = name

    def __call__(self, step):
        lr0 = float(self.initial_learning_rate)
        current_step = float(step)
        max_steps = float(self.decay_steps) - self.warmup_steps
        current_step = min(current_step, max_steps)
        slow = self.warmup_learning_rate
        adjusted = current_step
        if self.warmup_steps > 0:
            adjusted = max(adjusted - self.warmup_steps, 0.0)
            slow = self.warmup_learning_rate + (lr0 - self.warmup_learning_rate) * current_step / self.warmup_steps
        p = adjusted / max_steps
        lr = (lr0 - float(self.end_learning_rate)) * (1.0 - p) ** float(self.power) + float(self.end_learning_rate)
        return slow if step < self.warmup_steps else lr

    def get_config(self):
        return {"initial_learning_rate": self.initial_learning_rate, "decay_steps": self.decay_steps,
                "end_learning_rate": self.end_learning_rate, "power": self.power, "warmup_steps": self.warmup_steps,
                "warmup_learning_rate": self.warmup_learning_rate, "name": self.name}


class CosineDecay:
    """keras.optimizers.schedules.CosineDecay(initial, decay_steps, alpha, warmup_steps, warmup_target) (core_optimizer.py:147-160)"""

    def __init__(self, initial_learning_rate, decay_steps, alpha=0.0, warmup_steps=0, warmup_target=None):
        self.initial_learning_rate, self.decay_steps, self.alpha = initial_learning_rate, decay_steps, alpha
        self.warmup_steps, self.warmup_target = warmup_steps, warmup_target

    def __call__(self, step):
        import math

        if self.warmup_target is not None and step < self.warmup_steps:
            return self.initial_learning_rate + (self.warmup_target - self.initial_learning_rate) * step / self.warmup_steps
        peak = self.warmup_target if self.warmup_target is not None else self.initial_learning_rate
        s = min(max(step - (self.warmup_steps if self.warmup_target is not None else 0), 0), self.decay_steps)
        cosine = 0.5 * (1 + math.cos(math.pi * s / self.decay_steps))
        return peak * ((1 - self.alpha) * cosine + self.alpha)
