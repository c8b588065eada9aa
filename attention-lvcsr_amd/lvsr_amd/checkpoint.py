"""Parameter interchange in the reference's checkpoint format: a tar archive whose member `_parameters` is a
numpy .npz with one array per parameter, named by brick path with '|' as the delimiter
(libs/blocks/blocks/serialization.py:136,183-193,264-282,606-610).  Plain .npz files are accepted as well.
"""
import io
import os
import tarfile
import tempfile

import numpy

BRICK_DELIMITER = "|"


def load_parameters(path_or_file):
    """-> {'/recognizer/...': ndarray}  (serialization.py:264-282)."""
    if isinstance(path_or_file, (str, os.PathLike)):
        with open(path_or_file, "rb") as f:
            return load_parameters(f)
    data = path_or_file.read()
    bio = io.BytesIO(data)
    if tarfile.is_tarfile(bio):
        bio.seek(0)
        with tarfile.open(fileobj=bio, mode="r") as tar:
            member = tar.extractfile(tar.getmember("_parameters"))
            npz = numpy.load(io.BytesIO(member.read()), allow_pickle=False)
    else:
        bio.seek(0)
        npz = numpy.load(bio, allow_pickle=False)
    return {name.replace(BRICK_DELIMITER, "/"): npz[name] for name in npz.files}


def save_parameters(path, values):
    """Write `values` ({'/recognizer/...': ndarray}) as a Blocks-style tar with a `_parameters` member
    (serialization.py:183-193: numpy.savez with '|'-delimited names)."""
    with tempfile.NamedTemporaryFile("wb", suffix=".npz", delete=False) as tmp:
        numpy.savez(tmp, **{k.replace("/", BRICK_DELIMITER): numpy.asarray(v) for k, v in values.items()})
        tmp_name = tmp.name
    try:
        with tarfile.open(path, "w") as tar:
            tar.add(tmp_name, arcname="_parameters")
    finally:
        os.remove(tmp_name)
