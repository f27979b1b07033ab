"""Small tensor helpers of the WHVI path (mirror of the reference's src/utils.py)."""
import torch

__all__ = ["matmul_diag_left", "matmul_diag_right", "is_pow_of_2", "kl_normal", "kl_diag_normal",
           "build_H", "build_H_recursive"]


def matmul_diag_left(D_diagonal: torch.Tensor, A: torch.Tensor) -> torch.Tensor:
    """``diag(D_diagonal) @ A`` without forming the diagonal matrix: row i of A is scaled by
    ``D_diagonal[i]`` (src/utils.py:4-12).  One multiply (one rounding) per element."""
    return D_diagonal.unsqueeze(-1) * A


def matmul_diag_right(A: torch.Tensor, D_diagonal: torch.Tensor) -> torch.Tensor:
    """``A @ diag(D_diagonal)``: column j of A is scaled by ``D_diagonal[j]`` (src/utils.py:15-23)."""
    return A * D_diagonal


def is_pow_of_2(x) -> bool:
    """True for 1, 2, 4, ... (src/utils.py:26-33; falsy for 0)."""
    return bool(x) and (x & (x - 1)) == 0


def kl_normal(mu1, sd1, mu2, sd2):
    """Element-wise KL(N(mu1, sd1^2) || N(mu2, sd2^2)) with standard deviations
    (src/utils.py:36-46)."""
    return torch.log(sd2) - torch.log(sd1) + (sd1 ** 2 + (mu1 - mu2) ** 2) / (2 * sd2 ** 2) - 0.5


def kl_diag_normal(mu1, sd1, mu2, sd2):
    """KL between two diagonal Gaussians, in the reference's exact formula (src/utils.py:49-71).

    NOTE (kept on purpose): the formula treats ``sd1``/``sd2`` as VARIANCES
    (test/utils.py:29-33 checks it against ``MultivariateNormal(mu, diag(sd))``) although the
    caller passes a standard deviation for ``sd1`` (src/weights.py:59-64)."""
    assert mu1.size() == mu2.size() == sd1.size() == sd2.size()
    d = len(mu1)
    delta = mu2 - mu1
    return 0.5 * (torch.sum(torch.log(sd2)) - torch.sum(torch.log(sd1)) - d
                  + torch.sum(sd1 / sd2) + delta @ (delta / sd2))


def build_H_recursive(D: int) -> torch.Tensor:
    """Sylvester construction H_2n = [[H_n, H_n], [H_n, -H_n]] (src/utils.py:88-101), built
    iteratively by Kronecker doubling."""
    H = torch.tensor([[1.0]])
    n = 1
    while n < D:
        H = torch.cat([torch.cat([H, H], dim=1), torch.cat([H, -H], dim=1)], dim=0)
        n *= 2
    return H


def build_H(D: int, device) -> torch.Tensor:
    """Dense (D, D) Walsh-Hadamard matrix in natural order on ``device`` (src/utils.py:74-85)."""
    assert is_pow_of_2(D)
    return build_H_recursive(D).to(device)
