"""AvgMeter / get_lr, same behaviour as /root/reference/utils.py:1-21."""


class AvgMeter:
    """Sample-weighted running mean (utils.py:1-16)."""

    def __init__(self, name="Metric"):
        self.name = name
        self.reset()

    def reset(self):
        self.avg, self.sum, self.count = [0] * 3

    def update(self, val, count=1):
        self.count += count
        self.sum += val * count
        self.avg = self.sum / self.count

    def __repr__(self):
        return f"{self.name}: {self.avg:.4f}"


def get_lr(optimizer):
    """LR of the first param group (utils.py:19-21)."""
    for param_group in optimizer.param_groups:
        return param_group["lr"]
