"""MI355X-native SuperPoint + SuperGlue feature matching (drop-in for the hot path of
PH8411/image-matching).  Python here is host glue over the C-ABI library
``libimx.so`` (include/imx.h); all arithmetic on the product path runs in HIP kernels."""
__version__ = "0.4.0"      # = the "imx 0.4" of imx_version() (tests/test_host.py holds the two together)
