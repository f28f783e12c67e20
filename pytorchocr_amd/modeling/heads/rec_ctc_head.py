"""placeholder, filled in with the CRNN head."""
from torch import nn


class CTCHead(nn.Module):
    def __init__(self, in_channels, out_channels, **kwargs):
        raise NotImplementedError
