"""MI355X-native hot path of ebwt2InDel: packed-BWT rank/LF, LCP-flag construction, cluster scan and
left-context extension as hand-written HIP kernels behind a C ABI (include/ebwt2indel.h).

`ebwt2indel_amd.api` is the host-side mirror of the reference's interface; `ebwt2indel_amd.sim` makes
synthetic inputs.  Importing `api` requires the built HIP library -- there is no CPU fallback.
"""
__all__ = ["api", "sim", "_lib"]
