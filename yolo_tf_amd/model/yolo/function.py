"""model/yolo/function.py of the reference: ``leaky_relu(inputs, alpha=.1) = max(inputs, alpha * inputs)`` (:21-24).

On this path the activation never runs on its own: it is the epilogue of the convolution / fully connected kernel
(``yolo2_conv2d_bias_leaky``) or of the batch-norm kernels, and its gradient is ``yolo2_leaky_bwd`` / the BN backward kernels.
The symbol is kept because the inference plugins name it as their ``activation_fn``."""

ALPHA = 0.1


def leaky_relu(inputs, alpha=ALPHA):
    """Marks a layer's activation (the graph layer functions take ``activation=True``); on a NumPy array it evaluates the
    reference formula, which is what host-side callers (tests, tools) get."""
    import numpy as np
    a = np.asarray(inputs)
    return np.maximum(a, a.dtype.type(alpha) * a)
