"""MI355X-native render path behind the scene / camera / sampling surface of
bsati/raytracer-rust.  The compute path is the HIP library `librtamd.so`
(C ABI: include/rtamd.h, include/rtamd_host.h); this package is the thin host
mirror of the reference's interface plus torch plumbing (device memory, streams,
torch.distributed).  There is no CPU fallback: importing works without a GPU,
rendering fails loudly without the library or a gfx950 device."""
from .lib import RtError, load_library, library_path  # noqa: F401
from .api import (Context, RenderParams, SSAADecodeError, Scene, SuperSampling, accum_reduce, accum_reduce_rccl, compute_image, read_png, render_image_devices,  # noqa: F401
                  upload_devices, write_png, write_png16)
