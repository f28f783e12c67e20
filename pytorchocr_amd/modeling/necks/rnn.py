"""placeholder, filled in with the CRNN sequence encoder."""
from torch import nn


class SequenceEncoder(nn.Module):
    def __init__(self, in_channels, encoder_type, hidden_size=256, **kwargs):
        raise NotImplementedError
