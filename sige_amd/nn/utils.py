import torch

_ACTIVATIONS = {
    "relu": torch.relu,
    "sigmoid": torch.sigmoid,
    "tanh": torch.tanh,
    "swish": lambda x: x * torch.sigmoid(x),
    "identity": lambda x: x,
}


def activation(x: torch.Tensor, activation_name: str) -> torch.Tensor:
    """Elementwise activation used by the `profile`-mode dummies
    (API of sige/nn/utils.py:4-16)."""
    try:
        return _ACTIVATIONS[activation_name](x)
    except KeyError:
        raise ValueError("Unknown activation: [%s]!!!" % activation_name)
