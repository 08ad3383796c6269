"""torch.autograd glue: the bank's loss as a differentiable function of Q.

This is the seam a maintainer of the reference would use: in
cherryml/estimation/_ratelearn/trainer.py:170-177 the three torch calls
(matrix_exp, log, weighted sum) become `bank_loss(Q, pi, bank)`; everything
around it (parameterisation, Adam) stays torch."""
import torch


class _BankLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, pi, bank, normalize):
        loss, dQ = bank.loss_grad_torch(Q, pi, normalize=normalize, want_grad=True)
        ctx.save_for_backward(dQ)
        ctx.q_shape = Q.shape
        ctx.q_dtype = Q.dtype
        return loss.to(Q.dtype)

    @staticmethod
    def backward(ctx, grad_loss):
        (dQ,) = ctx.saved_tensors
        g = dQ * grad_loss.to(dQ.dtype).reshape(-1, 1, 1)
        return g.reshape(ctx.q_shape).to(ctx.q_dtype), None, None, None


def bank_loss(Q: torch.Tensor, pi: torch.Tensor, bank, normalize: bool = True) -> torch.Tensor:
    """loss[L] = -sum_b <C[l,b], log expm(t[l,b] Q[l])> (/ n_l); differentiable in Q.
    `pi` (the stationary distribution Q is reversible for) is used detached;
    pi=None selects the general scaling-and-squaring path (any rate matrix)."""
    return _BankLoss.apply(Q, None if pi is None else pi.detach(), bank, normalize)
