"""Matrix-free pieces of the reconstruction loop's normal-equation solve (NonObstacle.jl:351-356), for grids
where the reference's `MtM = M[order,Idx]' * M[order,Idx]; A \\ b` (CHOLMOD) cannot be formed: the action of MtM
comes from the device CSR (ray_frechet_normal_matvec_dev), the regulariser is applied as the stencil it is, and
the system is solved by Jacobi-preconditioned conjugate gradients on device vectors (torch tensors are the
caller's plumbing here; the O(nnz) work is the library's).  One all-reduce per CG iteration when the rays are
sharded over ranks."""
import torch


def regularizer_apply(u, N, h):
    """R u for R = Gx'Gx + Gy'Gy of regularization() (NonObstacle.jl:245-274), u a length-N^2 vector in node
    order i + j N: first differences towards i-1 and j+1 on the nodes i, j = 3..N-3 (1-based)."""
    U = u.view(N, N)                                   # U[j, i]
    out = torch.zeros_like(U)
    s = slice(2, N - 3)                                # 0-based 2..N-4
    g = (U[s, 2:N - 3] - U[s, 1:N - 4]) / (h * h)      # Gx: (node) - (node - 1), then Gx' scatters +g, -g
    out[s, 2:N - 3] += g
    out[s, 1:N - 4] -= g
    g = (U[2:N - 3, s] - U[3:N - 2, s]) / (h * h)      # Gy: (node) - (node + N)
    out[2:N - 3, s] += g
    out[3:N - 2, s] -= g
    return out.view(-1)


def pcg(apply_A, b, diag, rtol=1e-12, maxit=20000):
    """Jacobi-preconditioned CG for the SPD system A x = b; apply_A(p) -> A p.  Returns (x, iterations, relres)."""
    x = torch.zeros_like(b)
    r = b.clone()
    z = r / diag
    p = z.clone()
    rz = torch.dot(r, z)
    bnorm = float(torch.linalg.vector_norm(b))
    if bnorm == 0.0:
        return x, 0, 0.0
    res = 1.0
    for k in range(1, maxit + 1):
        Ap = apply_A(p)
        alpha = rz / torch.dot(p, Ap)
        x.add_(p, alpha=float(alpha))
        r.sub_(Ap, alpha=float(alpha))
        res = float(torch.linalg.vector_norm(r)) / bnorm
        if res <= rtol:
            return x, k, res
        z = r / diag
        rz_new = torch.dot(r, z)
        p.mul_(float(rz_new / rz)).add_(z)
        rz = rz_new
    return x, maxit, res
