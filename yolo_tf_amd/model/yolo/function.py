"""leaky_relu (reference model/yolo/function.py:21-24): max(x, 0.1*x).  On the MI355X path the
activation is never a standalone op: it is fused into the batch-norm apply kernel
(csrc/elementwise.hip bn_leaky_kernel) and its gradient into the BN backward kernels, so this
module only carries the constant the kernels are launched with."""
ALPHA = 0.1


def leaky_relu(inputs, alpha=ALPHA):
    raise NotImplementedError('leaky_relu is fused into graph.conv2d(..., activation=True); '
                              'it does not exist as a separate op on this path')
