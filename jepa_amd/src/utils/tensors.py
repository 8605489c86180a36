"""Host tensor helpers with the reference's names (src/utils/tensors.py)."""
import torch


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """Truncated normal by inverse-CDF on a uniform draw; same algorithm and RNG consumption as the reference's
    trunc_normal_ (src/utils/tensors.py:17-50), which is itself the torch.nn.init implementation."""
    return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def repeat_interleave_batch(x, B, repeat):
    """[x_0 .. x_{N-1}] blocks of B rows each -> every block repeated `repeat` times in place
    (src/utils/tensors.py:65-71; used for num_clips > 1)."""
    n = len(x) // B
    if repeat == 1:
        return x[:n * B]
    return x[:n * B].reshape(n, 1, B, *x.shape[1:]).expand(n, repeat, B, *x.shape[1:]).reshape(n * repeat * B, *x.shape[1:])
