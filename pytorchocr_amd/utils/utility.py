"""Host utilities on the path.  `sort_boxes` mirrors reference pytocr/utils/utility.py:32-50 (row P8 of SURVEY 8a)."""
import numpy as np


def sort_boxes(dt_boxes):
    """Sort text boxes top-to-bottom, left-to-right: stable sort by (y0, x0) of vertex 0, then ONE pass of adjacent
    swaps when two neighbours are within 10 px in y and out of order in x.  Arithmetic stays in the array's dtype
    (int16 from DBPostProcess), like the reference."""
    num_boxes = dt_boxes.shape[0]
    sorted_boxes = sorted(dt_boxes, key=lambda x: (x[0][1], x[0][0]))
    _boxes = list(sorted_boxes)
    for i in range(num_boxes - 1):
        if abs(_boxes[i + 1][0][1] - _boxes[i][0][1]) < 10 and (_boxes[i + 1][0][0] < _boxes[i][0][0]):
            tmp = _boxes[i]
            _boxes[i] = _boxes[i + 1]
            _boxes[i + 1] = tmp
    return _boxes
