"""DBPostProcess on the MI355X.

Mirror of reference `DBPostProcess` (pytocr/postprocess/db_postprocess.py:10-74) with cpp_speedup=True
semantics, i.e. the behaviour of the pybind11 extension db_postprocess_fast (src/db_postprocess.cpp:231-358):
hard-coded min_size 3 / max_candidates 1000, long-side size filter, polygon score on the raw contour with
4-connected edges, roundf rescale, scores reported as 1.0.  The whole batch runs on the device through
`ptocr_db_postprocess`; only int16 boxes and counts come back to the host.

`db_postprocess(pred, bitmap, ...)` below keeps the exact signature of the pybind function the reference
binds (src/db_postprocess.cpp:362-370) for callers that used the extension directly.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib

MAX_CANDIDATES = 1000


class _Workspace:
    def __init__(self):
        self.handle = None
        self.dims = (0, 0, 0)
        self.route = 0

    def set_route(self, route):
        """labelling route of the following calls: 0 = from the workspace's call history (default), 1 = text route, 2 = noise route
        (boxes never depend on it; kept across a regrowth of the workspace)"""
        self.route = int(route)
        if self.handle is not None:
            _lib.check(_lib.lib().ptocr_dbpost_set_route(self.handle, self.route), "ptocr_dbpost_set_route")

    def get(self, n, h, w):
        mn, mh, mw = self.dims
        if self.handle is None or n > mn or h > mh or w > mw or h * w > mh * mw:
            self.close()
            dims = (max(n, mn), max(h, mh), max(w, mw))
            hd = C.c_void_p()
            _lib.check(_lib.lib().ptocr_dbpost_create(C.byref(hd), *dims), "ptocr_dbpost_create")
            self.handle, self.dims = hd, dims
            if self.route:
                _lib.check(_lib.lib().ptocr_dbpost_set_route(hd, self.route), "ptocr_dbpost_set_route")
        return self.handle

    def host_buffers(self, n):
        """pinned host buffers for the results of an n-image call (pageable destinations make the runtime stage the 512 KB
        box array through dozens of small copy kernels)"""
        hb = getattr(self, "_host", None)
        if hb is None or hb[0].shape[0] < n:
            hb = (torch.empty((n, MAX_CANDIDATES, 4, 2), dtype=torch.int16, pin_memory=True),
                  torch.empty(n, dtype=torch.int32, pin_memory=True), torch.empty(n, dtype=torch.int32, pin_memory=True),
                  torch.empty((n, 2), dtype=torch.int32, pin_memory=True))
            self._host = hb
        return tuple(t[:n].numpy() for t in hb)

    def last_device_ms(self):
        """device time of the last call's kernels on this workspace (HIP events on the call's stream)"""
        ms = C.c_float(0)
        _lib.check(_lib.lib().ptocr_dbpost_last_device_ms(self.handle, C.byref(ms)), "ptocr_dbpost_last_device_ms")
        return float(ms.value)

    def close(self):
        if self.handle is not None:
            _lib.lib().ptocr_dbpost_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_ws = _Workspace()


def device_boxes(maps, src_wh, thresh, box_thresh, unclip_ratio, bitmap=None, use_padding_resize=False, ws=None, use_dilation=False):
    """maps: f32 cuda tensor [N,H,W]; src_wh: int array [N,2] (src_w, src_h); bitmap: optional u8 cuda tensor [N,H,W].
    -> (list of int16[K,4,2] per image, flags int32[N])"""
    if not maps.is_cuda:
        raise RuntimeError("device_boxes: maps must be on a cuda (ROCm) device; there is no CPU fallback")
    maps = maps.contiguous().float()
    n, h, w = maps.shape
    wsp = ws or _ws
    hd = wsp.get(n, h, w)
    boxes, counts, flags, src = wsp.host_buffers(n)
    src[...] = np.asarray(src_wh, np.int32).reshape(n, 2)
    counts[...] = 0
    flags[...] = 0
    bptr = C.c_void_p(0)
    if bitmap is not None:
        if not bitmap.is_cuda:
            raise RuntimeError("device_boxes: bitmap must be on the same cuda device")
        bitmap = bitmap.contiguous().to(torch.uint8)
        bptr = _lib.ptr(bitmap)
    _lib.check(_lib.lib().ptocr_db_postprocess_ex(
        hd, _lib.ptr(maps), bptr, n, h, w, C.c_float(thresh), C.c_float(box_thresh), C.c_float(unclip_ratio),
        src.ctypes.data_as(C.c_void_p), int(bool(use_padding_resize)), int(bool(use_dilation)), boxes.ctypes.data_as(C.c_void_p), MAX_CANDIDATES,
        counts.ctypes.data_as(C.c_void_p), flags.ctypes.data_as(C.c_void_p), _lib.cur_stream()), "ptocr_db_postprocess")
    return [boxes[i, :counts[i]].copy() for i in range(n)], flags.copy()


_FLAG_TEXT = {2: "a box score within 1e-6 of box_thresh was re-summed in the reference's raster order"}


def _warn_flags(flags):
    """The per-image exception flags of ptocr_db_postprocess are never silent: one warning per call that raised any."""
    import warnings
    bits = int(np.bitwise_or.reduce(np.asarray(flags, np.int64)))
    imgs = [int(i) for i in np.nonzero(flags)[0][:8]]
    warnings.warn("DBPostProcess: images %s%s raised exception flags: %s (see DBPostProcess.last_flags)" % (
        imgs, "..." if int(np.count_nonzero(flags)) > 8 else "", "; ".join(t for b, t in _FLAG_TEXT.items() if bits & b)),
        RuntimeWarning, stacklevel=3)


class DBPostProcess(object):
    """The post process for Differentiable Binarization (DB) -- same constructor and call contract as the reference."""

    def __init__(self, thresh=0.3, box_thresh=0.5, max_candidates=1000, unclip_ratio=1.5, use_dilation=False,
                 score_mode="poly", cpp_speedup=False, out_polygon=False, **kwargs):
        self.thresh = thresh
        self.box_thresh = box_thresh
        # kept for the contract only: the C++ path this class follows hard-codes 1000 (db_postprocess.cpp:238-239) and never
        # reads the Python attribute (db_postprocess.py:57-63 passes no max_candidates to cpp_boxes_from_bitmap)
        self.max_candidates = max_candidates
        self.unclip_ratio = unclip_ratio
        self.min_size = 3
        self.out_polygon = out_polygon
        self.score_mode = score_mode
        assert score_mode in ["box", "poly"], "Score mode must be in [box, poly] but got: {}".format(score_mode)
        self.use_dilation = use_dilation
        self.cpp_speedup = cpp_speedup
        self.last_flags = None
        self.device_ms_log = None        # bench.py sets this to a list: device ms of every call, in submission order
        self._ws = _Workspace()          # per-instance workspace: instances may run on different streams / threads
        self._pool = None
        self._stream = None
        if not cpp_speedup:
            raise NotImplementedError(
                "pytorchocr_amd DBPostProcess implements the cpp_speedup=True semantics (the reference's C++ extension, "
                "what configs/det/det_r18_db.yml runs); the pure-Python branch differs from it (short-side filter, "
                "np.round, 8-connected fill) and is not built")
        if out_polygon:
            raise NotImplementedError("out_polygon=True needs cpp_speedup=False in the reference; not on the hot path")

    def submit(self, outs_dict, shape_list, use_padding_resize=False):
        """Asynchronous __call__: returns a future whose .result() is the reference's return value.  The post-process
        runs on this instance's own HIP stream from a worker thread (the C call releases the GIL), ordered after the
        work already queued on the caller's current stream -- so the next batch's convolutions overlap with it."""
        import concurrent.futures
        pred = outs_dict["maps"]
        if isinstance(pred, np.ndarray):
            pred = torch.from_numpy(np.ascontiguousarray(pred, np.float32))
        if not pred.is_cuda:
            pred = pred.to("cuda:%d" % torch.cuda.current_device())
        dev = pred.device
        if self._pool is None:
            self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="dbpost")
            self._stream = torch.cuda.Stream(device=dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        pred.record_stream(self._stream)
        shape_list = np.asarray(shape_list).copy()

        def work():
            torch.cuda.set_device(dev)
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(ready)
                return self._run(pred, shape_list, use_padding_resize)
        return self._pool.submit(work)

    def __call__(self, outs_dict, shape_list, use_padding_resize=False):
        pred = outs_dict["maps"]
        if isinstance(pred, np.ndarray):
            pred = torch.from_numpy(np.ascontiguousarray(pred, np.float32))
        if not pred.is_cuda:
            # host buffers are accepted like the reference does, but the work still happens on the GPU
            pred = pred.to("cuda:%d" % torch.cuda.current_device())
        return self._run(pred, shape_list, use_padding_resize)

    def _run(self, pred, shape_list, use_padding_resize):
        pred = pred[:, 0, :, :]
        shape_list = np.asarray(shape_list)
        src_wh = np.stack([shape_list[:, 1].astype(np.int64), shape_list[:, 0].astype(np.int64)], axis=1)   # (src_w, src_h)
        boxes, flags = device_boxes(pred, src_wh, self.thresh, self.box_thresh, self.unclip_ratio,
                                    use_padding_resize=use_padding_resize, ws=self._ws, use_dilation=self.use_dilation)
        self.last_flags = flags
        if self.device_ms_log is not None:
            self.device_ms_log.append(self._ws.last_device_ms())
        if flags.any():
            _warn_flags(flags)
        return [{"points": b, "scores": [1.0] * len(b)} for b in boxes]


def db_postprocess(pred, bitmap, box_thresh, det_db_unclip_ratio, src_w, src_h, use_padding_resize=False):
    """Drop-in for the pybind11 `db_postprocess.db_postprocess` (reference src/db_postprocess.cpp:319-370):
    pred f32[H,W], bitmap u8[H,W] (host arrays) -> list[K][4][2] of int."""
    p = torch.from_numpy(np.ascontiguousarray(pred, np.float32))[None].cuda()
    b = torch.from_numpy(np.ascontiguousarray(bitmap).astype(np.uint8))[None].cuda()
    boxes, _ = device_boxes(p, [[src_w, src_h]], 0.0, box_thresh, det_db_unclip_ratio, bitmap=b,
                            use_padding_resize=use_padding_resize)
    return boxes[0].astype(np.int64).tolist()
