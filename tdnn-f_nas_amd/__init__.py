"""MI355X-native LF-MMI TDNN-F / DARTS hot path (drop-in for the nnet3 Component
surface of skhu101/TDNN-F_NAS).  The directory is named tdnn-f_nas_amd/; it is
imported under the alias `tdnnf_nas_amd` by __graft_entry__.load_package().

Python here is plumbing only (ctypes binding of the C-ABI in include/tdnnf_hip.h,
synthetic data, torch.distributed glue); the product is csrc/ (HIP kernels, C++
host mirror of the nnet3 components and the chain trainer step)."""
from . import synth  # noqa: F401
from . import hipabi  # noqa: F401
from . import trainer  # noqa: F401
from . import derive  # noqa: F401
from . import configs  # noqa: F401
from . import outer_loop  # noqa: F401
from . import egs  # noqa: F401
