"""placeholder, filled in with the CRNN decode."""


class CTCLabelDecode(object):
    def __init__(self, *a, **k):
        raise NotImplementedError
