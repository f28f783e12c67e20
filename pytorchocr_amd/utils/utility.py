"""Host utilities on the path.  `sort_boxes` has the semantics of reference pytocr/utils/utility.py:32-50 (row P8 of
SURVEY 8a); `get_part_img` (:53-78) lives in utils/warp.py and is re-exported here under the reference's module path."""
from .warp import get_part_img  # noqa: F401


def sort_boxes(dt_boxes):
    """Reading order for text boxes [K,4,2]: a STABLE sort on vertex 0 by (y, x), followed by exactly one left-to-right pass
    that swaps neighbours whose vertex-0 rows are closer than 10 px while their x order is inverted (one pass only: a box
    moves at most one place per earlier neighbour, as in the reference).  Returns a list of the K [4,2] arrays; comparisons
    and the |dy| run in the array's own dtype (int16 from DBPostProcess), like the reference."""
    import numpy as np
    if isinstance(dt_boxes, np.ndarray) and dt_boxes.dtype == np.int16 and dt_boxes.ndim == 3:
        # the same order computed on plain ints (numpy scalar indexing in the two loops cost 125 us per image of 170 boxes): lexsort is
        # stable like sorted(); |dy| wraps like int16 arithmetic does (abs(-32768) stays -32768)
        x0, y0 = dt_boxes[:, 0, 0], dt_boxes[:, 0, 1]
        order = np.lexsort((x0, y0)).tolist()
        xs, ys = x0.tolist(), y0.tolist()
        for i in range(1, len(order)):
            u, l = order[i - 1], order[i]
            d = ((ys[l] - ys[u] + 32768) & 0xFFFF) - 32768
            if (d if d == -32768 else abs(d)) < 10 and xs[l] < xs[u]:
                order[i - 1], order[i] = l, u
        return [dt_boxes[i] for i in order]
    order = sorted(range(dt_boxes.shape[0]), key=lambda i: (dt_boxes[i][0][1], dt_boxes[i][0][0]))
    boxes = [dt_boxes[i] for i in order]
    for i in range(1, len(boxes)):
        upper, lower = boxes[i - 1], boxes[i]
        if abs(lower[0][1] - upper[0][1]) < 10 and lower[0][0] < upper[0][0]:
            boxes[i - 1], boxes[i] = lower, upper
    return boxes
