"""Build recipe (TEST INFRASTRUCTURE): compiles the reference's Cython soft-NMS, src/lib/external/nms.pyx, from where it
lies under /root/reference into oracle/_ref/ (git-ignored; nothing of the reference is copied into the repository).

The file does not cythonize against this image's numpy 2.2 as it stands: its first function, `nms` (lines 24-75, NOT on
the polydet path), declares `np.ndarray[np.int_t, ...]` and calls `np.zeros(..., dtype=np.int)` -- the `int_t` typedef
and the `np.int` alias were removed from numpy (2.0 / 1.24).  The recipe therefore feeds Cython the text with exactly
two token substitutions applied in memory, both inside `nms()`:

    np.int_t  ->  np.intp_t        (argsort's index type on this platform, what `int_t` was: C long)
    np.int)   ->  np.intp)         (the removed alias of Python int as a dtype: C long)

`soft_nms` (lines 77-170), the function the detector calls (src/lib/detectors/polydet.py:66-67) and the only one the
fixtures exercise, contains neither token: it is compiled from the reference's unmodified text.  The patched text lives
only in a temporary directory that is deleted afterwards.

    python oracle/build_ref_nms.py        ->  oracle/_ref/refnms*.so   (module name `refnms`, function `soft_nms`)

Needs /root/reference (absent on the GPU box: the fixtures tests/golden/softnms_*.npz are what travels)."""
import glob
import os
import re
import shutil
import subprocess
import sys
import sysconfig
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/lib/external/nms.pyx"
OUT = os.path.join(ROOT, "oracle", "_ref")


def build():
    import numpy
    if not os.path.exists(SRC):
        raise FileNotFoundError(SRC)
    os.makedirs(OUT, exist_ok=True)
    text = open(SRC).read()
    body = text[text.index("def soft_nms("):text.index("def soft_nms_39(")]
    assert "np.int" not in body, "soft_nms itself would be patched: the pin would not be the reference's text"
    text, n1 = re.subn(r"np\.int_t\b", "np.intp_t", text)
    text, n2 = re.subn(r"np\.int\)", "np.intp)", text)
    assert (n1, n2) == (2, 1), (n1, n2)
    tmp = tempfile.mkdtemp(prefix="refnms_")
    try:
        pyx = os.path.join(tmp, "refnms.pyx")
        with open(pyx, "w") as fh:
            fh.write(text)
        subprocess.run([sys.executable, "-m", "cython", "-2", pyx, "-o", os.path.join(tmp, "refnms.c")], check=True)
        so = os.path.join(OUT, "refnms" + sysconfig.get_config_var("EXT_SUFFIX"))
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-Wno-cpp", "-Wno-unused-function",
                        "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
                        "-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include(),
                        os.path.join(tmp, "refnms.c"), "-o", so], check=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return so


def load():
    """Import the built module (building it first when /root/reference is present and it is missing)."""
    if not glob.glob(os.path.join(OUT, "refnms*.so")):
        build()
    if OUT not in sys.path:
        sys.path.insert(0, OUT)
    import refnms
    return refnms


if __name__ == "__main__":
    print(build())
