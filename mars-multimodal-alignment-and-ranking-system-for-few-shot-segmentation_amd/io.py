"""Files either side of the ranking path: stored proposals, checkpoints, split lists.

* proposals: the reference reads one ``{fold}_{idx}.pt`` float tensor [N, H, W] per episode
  (/root/reference/main_MARS.py:62).  ``load_proposals`` reads those with ``weights_only=True`` and also a
  bit-packed form (``.mprop``: 32x smaller on disk and over PCIe, unpacked on the GPU).
* checkpoints: plain state dicts (optionally wrapped in ``{"model": ...}`` / ``{"state_dict": ...}``, optionally
  with a ``module.`` prefix), read with ``weights_only=True``; of TorchScript archives only the tensors are read (no code of the archive runs), nothing in a
  file is ever executed (/root/reference/utils/backbone_loader.py:17-52, segment_anything/build_sam.py:103-106).
* split lists: ``read_plain_pickle`` rebuilds containers of plain values from a pickle's opcode listing and
  refuses every opcode that would import or call something, for split files like the reference's
  ``datasets/COCO2014/splits/*/fold*.pkl`` (class id -> list of image names).
"""
from __future__ import annotations

import pickletools
import struct
import zipfile
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import ops

MAGIC = b"MPRP"
_HEADER = struct.Struct("<4sIIII")          # magic, version, n, h, w


# ------------------------------------------------------------------ proposals
def save_packed(path: str, masks: torch.Tensor) -> None:
    """masks [N, H, W] (GPU, any dtype, > 0 = set) -> ``.mprop`` file."""
    m = masks.contiguous().float()
    n, h, w = m.shape
    words = torch.empty(n, (h * w + 31) // 32, dtype=torch.int32, device=m.device)
    ops.pack_masks(m, words)
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, 1, n, h, w))
        f.write(words.cpu().numpy().tobytes())


def load_packed(path: str, device) -> torch.Tensor:
    with open(path, "rb") as f:
        magic, version, n, h, w = _HEADER.unpack(f.read(_HEADER.size))
        if magic != MAGIC or version != 1:
            raise ValueError(f"{path}: not a packed proposal file")
        nwords = (h * w + 31) // 32
        raw = f.read()
    if len(raw) != 4 * n * nwords:
        raise ValueError(f"{path}: {len(raw)} payload bytes, expected {4 * n * nwords}")
    words = torch.from_numpy(np.frombuffer(raw, dtype=np.int32).reshape(n, nwords).copy()).to(device)
    masks = torch.empty(n, h, w, dtype=torch.float32, device=device)
    ops.unpack_masks(words, masks)
    return masks


def load_proposals(path: str, device) -> torch.Tensor:
    """[N, H, W] float {0, 1} on ``device`` from a ``.mprop`` file or a tensor file of the reference's kind."""
    if path.endswith(".mprop"):
        return load_packed(path, device)
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict):
        obj = obj.get("masks", next((v for v in obj.values() if torch.is_tensor(v)), None))
    if not torch.is_tensor(obj) or obj.dim() not in (3, 4):
        raise ValueError(f"{path}: expected a [N, H, W] tensor of proposals")
    if obj.dim() == 4:
        obj = obj[:, 0]
    return (obj.to(device) > 0).float().contiguous()


# ------------------------------------------------------------------ checkpoints
def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            scripted = any(n.endswith("constants.pkl") or "/code/" in n for n in z.namelist())
        if scripted:
            # the archives OpenAI publishes CLIP in (clip/clip.py:95-150 loads them with torch.jit.load and takes
            # state_dict()): the tensors are taken out of the archive's data pickle without running any of it
            return read_torchscript_state_dict(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("model", "state_dict", "model_state_dict"):
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
    if not isinstance(sd, dict) or not all(torch.is_tensor(v) for v in sd.values()):
        raise ValueError(f"{path}: not a state dict")
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


# ------------------------------------------------------------------ plain pickles
_MARK = object()


def read_plain_pickle(path: str, max_bytes: int = 1 << 30) -> Any:
    """Containers (dict, list, tuple, set) of None / bool / int / float / str / bytes from a pickle, built from
    ``pickletools.genops`` (a disassembler: it constructs nothing).  Any opcode that names a global, calls,
    builds an object or reads a persistent id raises ``ValueError``; nothing from the file is executed."""
    with open(path, "rb") as f:
        data = f.read(max_bytes + 1)
    if len(data) > max_bytes:
        raise ValueError(f"{path}: larger than {max_bytes} bytes")
    return _plain_objects(data, path, None)


def _plain_objects(data: bytes, path: str, extra) -> Any:
    """The object a pickle describes, from its opcode listing.  ``extra(name, arg, stack, pop_mark)`` may take the
    opcodes that are outside plain data (it returns False for what it does not take: those are refused)."""
    stack, memo = [], {}

    def pop_mark():
        i = len(stack) - 1
        while stack[i] is not _MARK:
            i -= 1
        items = stack[i + 1:]
        del stack[i:]
        return items

    for op, arg, _ in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        if n == "STOP":
            return stack.pop()
        if n == "MARK":
            stack.append(_MARK)
        elif n == "NONE":
            stack.append(None)
        elif n in ("NEWTRUE", "NEWFALSE"):
            stack.append(n == "NEWTRUE")
        elif n in ("INT", "BININT", "BININT1", "BININT2", "LONG", "LONG1", "LONG4", "FLOAT", "BINFLOAT", "STRING", "BINSTRING",
                   "SHORT_BINSTRING", "BINBYTES", "SHORT_BINBYTES", "BINBYTES8", "UNICODE", "BINUNICODE", "SHORT_BINUNICODE",
                   "BINUNICODE8"):
            stack.append(arg)
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "EMPTY_SET":
            stack.append(set())
        elif n == "DICT":
            it = pop_mark()
            stack.append(dict(zip(it[0::2], it[1::2])))
        elif n == "LIST":
            stack.append(pop_mark())
        elif n == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif n in ("TUPLE1", "TUPLE2", "TUPLE3"):
            k = int(n[-1])
            items = stack[-k:]
            del stack[-k:]
            stack.append(tuple(items))
        elif n == "FROZENSET":
            stack.append(frozenset(pop_mark()))
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n == "APPENDS":
            it = pop_mark()
            stack[-1].extend(it)
        elif n == "SETITEM":
            v, k = stack.pop(), stack.pop()
            stack[-1][k] = v
        elif n == "SETITEMS":
            it = pop_mark()
            stack[-1].update(zip(it[0::2], it[1::2]))
        elif n == "ADDITEMS":
            it = pop_mark()
            stack[-1].update(it)
        elif n in ("PUT", "BINPUT", "LONG_BINPUT"):
            memo[arg] = stack[-1]
        elif n == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif n in ("GET", "BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif n == "POP":
            stack.pop()
        elif n == "POP_MARK":
            pop_mark()
        elif n == "DUP":
            stack.append(stack[-1])
        elif extra is not None and extra(n, arg, stack, pop_mark):
            pass
        else:
            raise ValueError(f"{path}: opcode {n} is outside plain data (it would import or construct an object); refused")
    raise ValueError(f"{path}: truncated pickle")


# ------------------------------------------------------------------ TorchScript archives, tensors only
class _Named:
    """A global a pickle names (never imported), or an instance of one with the state the pickle gives it."""

    def __init__(self, module: str, name: str, instance: bool = False):
        self.module, self.name, self.instance, self.state = module, name, instance, None


_STORAGE_DTYPES = {
    "FloatStorage": torch.float32, "HalfStorage": torch.float16, "BFloat16Storage": torch.bfloat16, "DoubleStorage": torch.float64,
    "LongStorage": torch.int64, "IntStorage": torch.int32, "ShortStorage": torch.int16, "CharStorage": torch.int8,
    "ByteStorage": torch.uint8, "BoolStorage": torch.bool,
}


def read_torchscript_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """``torch.jit.load(path).state_dict()`` without loading the module: the archive's ``data.pkl`` is walked opcode by
    opcode (``pickletools.genops``), the classes it names stay names, and the only calls honoured are the ones
    written out here - ``torch._utils._rebuild_tensor_v2`` on a storage record of the archive (a strided view of the
    bytes of ``data/<key>``) and an empty ``collections.OrderedDict``.  Attribute paths of the module tree become
    the keys; anything else the pickle would import or call raises ``ValueError``."""
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        pkl = [n for n in names if n.endswith("/data.pkl") and n.count("/") == 1]
        if len(pkl) != 1:
            raise ValueError(f"{path}: not a TorchScript archive (no data.pkl)")
        root = pkl[0][: -len("data.pkl")]
        if root + "byteorder" in names and z.read(root + "byteorder").strip() not in (b"little", b""):
            raise ValueError(f"{path}: big-endian archive")
        storages: Dict[str, torch.Tensor] = {}

        def storage(rec) -> torch.Tensor:
            if not (isinstance(rec, tuple) and len(rec) == 5 and rec[0] == "storage" and isinstance(rec[1], _Named)
                    and rec[1].module == "torch" and rec[1].name in _STORAGE_DTYPES and isinstance(rec[2], str)):
                raise ValueError(f"{path}: persistent id {rec!r} is not a tensor storage; refused")
            key, dtype, numel = rec[2], _STORAGE_DTYPES[rec[1].name], int(rec[4])
            if key not in storages:
                name = root + "data/" + key
                if name not in names:
                    raise ValueError(f"{path}: storage {key} is missing from the archive")
                raw = bytearray(z.read(name))
                if len(raw) < numel * torch.empty(0, dtype=dtype).element_size():
                    raise ValueError(f"{path}: storage {key} is shorter than its record")
                storages[key] = torch.frombuffer(raw, dtype=dtype, count=numel) if numel else torch.empty(0, dtype=dtype)
            return storages[key]

        def extra(n, arg, stack, pop_mark) -> bool:
            if n == "GLOBAL":
                module, name = arg.split(" ", 1)
                stack.append(_Named(module, name))
            elif n == "STACK_GLOBAL":
                name, module = stack.pop(), stack.pop()
                stack.append(_Named(module, name))
            elif n == "BINPERSID":
                stack.append(storage(stack.pop()))
            elif n == "NEWOBJ":
                args, cls = stack.pop(), stack.pop()
                if not (isinstance(cls, _Named) and not cls.instance and cls.module.split(".")[0] == "__torch__" and args == ()):
                    raise ValueError(f"{path}: object of {getattr(cls, 'module', cls)!r} is not a scripted class; refused")
                stack.append(_Named(cls.module, cls.name, instance=True))
            elif n == "BUILD":
                state = stack.pop()
                if not (isinstance(stack[-1], _Named) and stack[-1].instance):
                    raise ValueError(f"{path}: state for something that is not a scripted object; refused")
                stack[-1].state = state
            elif n == "REDUCE":
                args, fn = stack.pop(), stack.pop()
                what = (fn.module, fn.name) if isinstance(fn, _Named) and not fn.instance else None
                if what == ("collections", "OrderedDict") and args == ():
                    stack.append({})
                elif what == ("torch._utils", "_rebuild_tensor_v2") and len(args) >= 4 and torch.is_tensor(args[0]):
                    st, offset, size, stride = args[0], int(args[1]), tuple(int(v) for v in args[2]), tuple(int(v) for v in args[3])
                    need = offset + sum((d - 1) * k for d, k in zip(size, stride)) + 1 if all(d > 0 for d in size) else 0
                    if offset < 0 or any(k < 0 for k in stride) or need > st.numel():
                        raise ValueError(f"{path}: tensor view outside its storage; refused")
                    stack.append(torch.as_strided(st, size, stride, offset))
                else:
                    raise ValueError(f"{path}: call of {what or fn!r} is not part of a tensor record; refused")
            else:
                return False
            return True

        module = _plain_objects(z.read(pkl[0]), path, extra)
    out: Dict[str, torch.Tensor] = {}

    def walk(obj, prefix: str):
        if torch.is_tensor(obj):
            out[prefix[:-1]] = obj.clone()
        elif isinstance(obj, _Named) and obj.instance and isinstance(obj.state, dict):
            for k, v in obj.state.items():
                if isinstance(k, str):
                    walk(v, prefix + k + ".")

    walk(module, "")
    if not out:
        raise ValueError(f"{path}: no tensors found in the archive's module")
    return out
