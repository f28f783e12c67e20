"""`pytocr` -- the reference's package name, served by pytorchocr_amd (the reference's deploy scripts import
`pytocr.data.{create_operators, transform}`, `pytocr.modeling.architectures.build_model`,
`pytocr.postprocess.build_post_process`, `pytocr.utils.save_load.load_pretrained_params` and
`pytocr.utils.utility.{sort_boxes, get_part_img}`; deploy/pytorch/infer_det.py:18-22, run_ocr.py:18-22).

Every `pytocr.X` import resolves to THE SAME module object as `pytorchocr_amd.X` (an alias, not a second copy: the HIP
library handle, packed weights and module-level switches exist once), so those scripts run on the MI355X path with their
import lines unchanged.  Sub-packages of the reference that are outside the inference hot path (losses, optimizer, training
datasets) do not exist here and raise ModuleNotFoundError as usual.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import pytorchocr_amd as _impl

_ALIAS, _REAL = "pytocr", "pytorchocr_amd"


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name):
        self.real_name = real_name

    def create_module(self, spec):
        mod = importlib.import_module(self.real_name)        # the already-imported (or now imported) real module
        self.real_spec = mod.__spec__
        return mod

    def exec_module(self, module):                            # nothing to run: the real module is initialised;
        module.__spec__ = self.real_spec                      # the import machinery re-pointed __spec__ at the alias: undo


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_ALIAS + "."):
            return None
        real_name = _REAL + fullname[len(_ALIAS):]
        try:
            real_spec = importlib.util.find_spec(real_name)
        except ModuleNotFoundError:
            return None
        if real_spec is None:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real_name), is_package=real_spec.submodule_search_locations is not None)


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

__path__ = []                # sub-modules come from the finder above, never from a directory scan


def __getattr__(name):       # pytocr.modeling etc. as attributes after `import pytocr`
    try:
        return importlib.import_module(_ALIAS + "." + name)
    except ModuleNotFoundError:
        raise AttributeError(name) from None
