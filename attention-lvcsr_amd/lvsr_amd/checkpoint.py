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


def save_parameters(path, values, extra=None):
    """Write `values` ({'/recognizer/...': ndarray}) as a Blocks-style tar with a `_parameters` member
    (serialization.py:183-193: numpy.savez with '|'-delimited names).  `extra`: {member name: {array name: ndarray}} is
    stored as further .npz members of the same tar (the reference pickles its whole main loop next to `_parameters`,
    serialization.py:145-262; here the training state a restart needs travels as plain arrays: `_training_state`)."""
    members = [("_parameters", {k.replace("/", BRICK_DELIMITER): numpy.asarray(v) for k, v in values.items()})]
    for name, arrays in (extra or {}).items():
        assert name != "_parameters"
        members.append((name, {k: numpy.asarray(v) for k, v in arrays.items()}))
    tmp_names = []
    try:
        for _, arrays in members:
            with tempfile.NamedTemporaryFile("wb", suffix=".npz", delete=False) as tmp:
                numpy.savez(tmp, **arrays)
                tmp_names.append(tmp.name)
        with tarfile.open(path, "w") as tar:
            for (name, _), tmp_name in zip(members, tmp_names):
                tar.add(tmp_name, arcname=name)
    finally:
        for tmp_name in tmp_names:
            os.remove(tmp_name)


def load_member(path, name):
    """-> {array name: ndarray} of an extra .npz member written by `save_parameters(extra=)`, or None when the archive has
    no such member (a checkpoint of the reference, or of an older run)."""
    with open(path, "rb") as f:
        bio = io.BytesIO(f.read())
    if not tarfile.is_tarfile(bio):
        return None
    bio.seek(0)
    with tarfile.open(fileobj=bio, mode="r") as tar:
        try:
            member = tar.extractfile(tar.getmember(name))
        except KeyError:
            return None
        npz = numpy.load(io.BytesIO(member.read()), allow_pickle=False)
        return {k: npz[k] for k in npz.files}
