"""BPE tokenizer against the reference tokenizer (CPU; needs the reference's merge table, so the
comparison runs only where /root/reference exists) and against the committed prompt ids."""
import importlib.util
import json
import os
import random
import sys
import types

import pytest
import torch

BPE = "/root/reference/clip/bpe_simple_vocab_16e6.txt.gz"
needs_table = pytest.mark.skipif(not os.path.exists(BPE), reason="CLIP merge table not available")


def _reference_tokenizer():
    if "ftfy" not in sys.modules:
        stub = types.ModuleType("ftfy")
        stub.fix_text = lambda s: s
        sys.modules["ftfy"] = stub
    spec = importlib.util.spec_from_file_location("ref_simple_tokenizer", "/root/reference/clip/simple_tokenizer.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.SimpleTokenizer(BPE)


@needs_table
def test_matches_reference_encoder(pkg):
    tok = pkg.tokenizer.BpeTokenizer(BPE)
    ref = _reference_tokenizer()
    rng = random.Random(0)
    words = ["a", "photo", "of", "without", "object", "person", "giraffe", "traffic", "light", "skateboard", "it's", "don't",
             "hair-drier", "24", "7up", "x", "zzzzqq", "antidisestablishmentarianism", "&amp;", "  spaced   out ", "UPPER", "né"]
    prompts = ["a object.", "a photo of object.", "a photo without object.", "a dog, a small brown animal with floppy ears.",
               "A  Cat,\tsitting&nbsp;on the mat!!", "<|startoftext|> odd <|endoftext|>", "the year 1999 and 3.14159", "he'll've"]
    for _ in range(200):
        prompts.append(" ".join(rng.choice(words) for _ in range(rng.randint(1, 9))) + rng.choice([".", "", "?!", " ..."]))
    for _ in range(100):
        prompts.append("".join(rng.choice("abcdefghijklmnopqrstuvwxyz ABC 0123456789.,;:'\"-_()[]{}!?@#$%^&*+=/<>|~`") for _ in range(rng.randint(1, 40))))
    for p in prompts:
        assert tok.encode(p) == ref.encode(p), p
    assert tok.sot == ref.encoder["<|startoftext|>"] and tok.eot == ref.encoder["<|endoftext|>"]
    assert len(tok.encoder) == len(ref.encoder) == 49408


@needs_table
def test_tokenize_shapes_padding_and_errors(pkg):
    tok = pkg.tokenizer.BpeTokenizer(BPE)
    out = tok(["a object.", "a photo of object."])
    assert out.dtype == torch.int32 and tuple(out.shape) == (2, 77)
    assert int(out[0, 0]) == tok.sot and int(out[0].argmax()) == int((out[0] != 0).sum()) - 1   # EOT is the largest id, last
    long_prompt = "word " * 100
    with pytest.raises(RuntimeError):
        tok(long_prompt)
    t = tok(long_prompt, truncate=True)
    assert int(t[0, -1]) == tok.eot and int(t[0, 0]) == tok.sot
    assert tok.decode(tok.encode("a photo of a giraffe.")).strip() == "a photo of a giraffe ."


@needs_table
def test_matches_committed_prompt_ids(pkg):
    tok = pkg.tokenizer.BpeTokenizer(BPE)
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prompt_tokens.json")))
    for prompt, ids in golden.items():
        assert tok(prompt)[0, : len(ids)].tolist() == ids, prompt
        if prompt in pkg.synth.PROMPT_TOKENS:
            assert torch.equal(tok(prompt), pkg.synth.tokenize(prompt))


def test_missing_table_is_reported(pkg, monkeypatch):
    monkeypatch.delenv("MARS_BPE_PATH", raising=False)
    with pytest.raises(FileNotFoundError):
        pkg.tokenizer.BpeTokenizer("/nonexistent/bpe.txt.gz")
    fallback = pkg.tokenizer.default_tokenizer("/nonexistent/bpe.txt.gz")
    assert tuple(fallback("a object.").shape) == (1, 77)
