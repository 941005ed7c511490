"""Transformer encoders behind the three stages, on PyTorch-ROCm.

The reference reaches its encoders through sentence-transformers
(``SentenceTransformer.encode`` at reference src/stage1_retriever.py:236-248 and
benchmark/tristage_mteb_model.py:187-193, ``CrossEncoder.predict`` at
src/stage3_reranker.py:127-131).  That package is a third-party dependency
(requirements.txt:3, ">=2.0.0", unpinned) and is not installed here, so its
published behaviour is restated on top of ``transformers``:

SentenceEncoder.encode      tokenise (pad to longest, truncate to max_seq_length)
                            -> AutoModel -> pooling module (mean over the
                            attention mask by default; cls / max) -> optional Dense
                            layers -> optional L2 normalisation; inputs are
                            processed in length-sorted batches and returned in
                            the caller's order, float32.
CrossEncoderModel.predict   tokenise (query, doc) pairs -> AutoModelForSequence-
                            Classification -> logits -> activation (Sigmoid for a
                            single label unless the checkpoint's config names
                            another one) -> float32 scores.

Model sources, in order: a local directory (``<cache_dir>/<basename>`` first, then
``<cache_dir>/<name>``, the reference's lookup at src/stage1_retriever.py:148-151),
else the hub name (needs network; never attempted in tests), or the offline spec
``random:<arch>[:hidden[:layers[:heads]]]`` which builds a randomly initialised
model of that architecture with a deterministic hashing tokenizer — used for the
throughput benchmarks and tests, where no weights are available ("parity
unpinned" for real checkpoints: no weights exist on disk).
"""
from __future__ import annotations

import json
import os
import re
import zlib
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F


def resolve_device(device: str = "auto") -> str:
    if device == "auto":
        return "cuda" if torch.cuda.is_available() else "cpu"
    return device


def local_model_dir(model_name: str, cache_dir: str) -> Optional[str]:
    """reference src/stage1_retriever.py:148-151 (same rule in stage 2 and 3)."""
    base = os.path.join(cache_dir, os.path.basename(model_name))
    legacy = os.path.join(cache_dir, model_name)
    if os.path.isdir(base):
        return base
    if os.path.isdir(legacy):
        return legacy
    if os.path.isdir(model_name):
        return model_name
    return None


# --------------------------------------------------------------------------- tokenizer
class HashTokenizer:
    """Deterministic offline tokenizer: lower-cased word / punctuation pieces hashed
    into a fixed vocabulary, BERT-style special tokens ([CLS]=101, [SEP]=102, [PAD]=0).
    Same call surface as the HF tokenizers for the arguments the reference uses
    (text, text_pair, truncation, padding, max_length, return_tensors="pt")."""

    pad_token_id, cls_token_id, sep_token_id = 0, 101, 102

    def __init__(self, vocab_size: int = 30522, model_max_length: int = 512):
        self.vocab_size = int(vocab_size)
        self.model_max_length = int(model_max_length)
        self._memo = {}

    def _ids_np(self, text: str) -> np.ndarray:
        hit = self._memo.get(text)          # the same documents / queries recur across pairs and queries
        if hit is not None:
            return hit
        pieces = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", text.lower())
        span = self.vocab_size - 1000
        ids = np.fromiter((1000 + (zlib.crc32(p.encode("utf-8")) % span) for p in pieces), dtype=np.int64, count=len(pieces))
        if len(self._memo) >= 100_000:
            self._memo.clear()
        self._memo[text] = ids
        return ids

    def _ids(self, text: str) -> List[int]:
        return self._ids_np(text).tolist()

    def __call__(self, text, text_pair=None, truncation=True, padding=False, max_length=None,
                 return_tensors=None, **_):
        single = isinstance(text, str)
        texts = [text] if single else list(text)
        pairs = None
        if text_pair is not None:
            pairs = [text_pair] if isinstance(text_pair, str) else list(text_pair)
        max_length = int(max_length or self.model_max_length)
        # per row: (first text ids, second text ids or None) after truncation — arrays, no per-token Python
        parts = []
        for i, t in enumerate(texts):
            a = self._ids_np(t)
            if pairs is None:
                if truncation:
                    a = a[: max(max_length - 2, 0)]
                parts.append((a, None))
            else:
                b = self._ids_np(pairs[i])
                if truncation:  # longest-first truncation, like HF's default strategy: one token at a
                    # time from the longer side, from the pair's second text on ties (closed form)
                    la, lb = len(a), len(b)
                    e = la + lb - max(max_length - 3, 0)
                    if e > 0:
                        if la > lb:
                            c = min(e, la - lb)
                            la, e = la - c, e - c
                        elif lb > la:
                            c = min(e, lb - la)
                            lb, e = lb - c, e - c
                        lb, la = lb - (e + 1) // 2, la - e // 2
                        a, b = a[:max(la, 0)], b[:max(lb, 0)]
                parts.append((a, b))
        lens = [len(a) + 2 + (len(b) + 1 if b is not None else 0) for a, b in parts]
        tensorise = padding or return_tensors == "pt"
        if tensorise and not padding and len(set(lens)) > 1:
            raise ValueError("cannot tensorise ragged rows without padding")
        if not tensorise:
            rows, types = [], []
            for a, b in parts:
                ids = [self.cls_token_id] + a.tolist() + [self.sep_token_id]
                tt = [0] * len(ids)
                if b is not None:
                    ids += b.tolist() + [self.sep_token_id]
                    tt += [1] * (len(b) + 1)
                rows.append(ids)
                types.append(tt)
            return {"input_ids": rows, "attention_mask": [[1] * len(r) for r in rows], "token_type_ids": types}
        n, width = len(parts), max(lens)
        ids = np.full((n, width), self.pad_token_id, dtype=np.int64)
        mask = np.zeros((n, width), dtype=np.int64)
        types = np.zeros((n, width), dtype=np.int64)
        for i, (a, b) in enumerate(parts):
            la = len(a)
            ids[i, 0] = self.cls_token_id
            ids[i, 1: 1 + la] = a
            ids[i, 1 + la] = self.sep_token_id
            if b is not None:
                lb = len(b)
                ids[i, 2 + la: 2 + la + lb] = b
                ids[i, 2 + la + lb] = self.sep_token_id
                types[i, 2 + la: 3 + la + lb] = 1
            mask[i, : lens[i]] = 1
        out = {"input_ids": ids, "attention_mask": mask, "token_type_ids": types}
        if return_tensors == "pt":
            return {k: torch.from_numpy(v) for k, v in out.items()}
        return {k: v.tolist() for k, v in out.items()}


# --------------------------------------------------------------------------- random models
_RANDOM_ARCHS = {
    # arch: (config class name, defaults) — shapes of the models BASELINE.json names
    "minilm": dict(kind="bert", hidden=384, layers=6, heads=12, inter=1536),          # all-MiniLM-L6 / ms-marco-MiniLM-L6
    "bert": dict(kind="bert", hidden=768, layers=12, heads=12, inter=3072),
    "modernbert": dict(kind="modernbert", hidden=768, layers=22, heads=12, inter=1152),  # GTE-ModernColBERT-v1 backbone
    "xlmr-large": dict(kind="xlm-roberta", hidden=1024, layers=24, heads=16, inter=4096),  # bge-reranker-large
    "tiny": dict(kind="bert", hidden=64, layers=2, heads=4, inter=128),
}


def _random_config(spec: str, num_labels: Optional[int] = None):
    import transformers
    parts = spec.split(":")
    arch = parts[1] if len(parts) > 1 and parts[1] else "tiny"
    if arch not in _RANDOM_ARCHS:
        raise ValueError(f"unknown random architecture {arch!r}; known: {sorted(_RANDOM_ARCHS)}")
    a = dict(_RANDOM_ARCHS[arch])
    if len(parts) > 2:
        a["hidden"] = int(parts[2])
        a["inter"] = 4 * a["hidden"]
    if len(parts) > 3:
        a["layers"] = int(parts[3])
    if len(parts) > 4:
        a["heads"] = int(parts[4])
    kw = dict(hidden_size=a["hidden"], num_hidden_layers=a["layers"], num_attention_heads=a["heads"],
              intermediate_size=a["inter"], vocab_size=30522, max_position_embeddings=514)
    if num_labels is not None:
        kw["num_labels"] = num_labels
    if a["kind"] == "bert":
        return transformers.BertConfig(**kw)
    if a["kind"] == "xlm-roberta":
        return transformers.XLMRobertaConfig(**kw, pad_token_id=0, type_vocab_size=2)
    if a["kind"] == "modernbert":
        kw.pop("max_position_embeddings")
        return transformers.ModernBertConfig(**kw, max_position_embeddings=8192, pad_token_id=0,
                                             cls_token_id=101, sep_token_id=102, bos_token_id=101,
                                             eos_token_id=102)
    raise ValueError(a["kind"])


def load_backbone(model_name: str, cache_dir: str = "./models", head: str = "base",
                  num_labels: int = 1, seed: int = 0):
    """Returns (tokenizer, model, source_dir_or_None).  head: 'base' (AutoModel) or
    'seqcls' (AutoModelForSequenceClassification)."""
    import transformers
    if model_name.startswith("random:"):
        cfg = _random_config(model_name, num_labels if head == "seqcls" else None)
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        try:
            model = (transformers.AutoModelForSequenceClassification.from_config(cfg) if head == "seqcls"
                     else transformers.AutoModel.from_config(cfg))
        finally:
            torch.random.set_rng_state(gen_state)
        return HashTokenizer(cfg.vocab_size), model, None
    src = local_model_dir(model_name, cache_dir) or model_name
    tok = transformers.AutoTokenizer.from_pretrained(src, cache_dir=cache_dir)
    cls = transformers.AutoModelForSequenceClassification if head == "seqcls" else transformers.AutoModel
    model = cls.from_pretrained(src, cache_dir=cache_dir)
    return tok, model, (src if os.path.isdir(src) else None)


def _autocast(device: str, enabled: bool, dtype=torch.bfloat16):
    if enabled and str(device).startswith("cuda"):
        return torch.autocast("cuda", dtype=dtype)
    import contextlib
    return contextlib.nullcontext()


# --------------------------------------------------------------------------- hipGraph replay
def to_device_async(t: torch.Tensor, device) -> torch.Tensor:
    """Host tensor -> GPU through pinned memory without blocking the host: a plain ``.to(device)`` of pageable memory
    waits for everything already queued on the stream (the previous batch's forward or scan), which serialises the host's
    tokenising and launching with the GPU's work.  (torch's pinned-memory cache keeps the staging buffer alive until
    the copy has run.)  Anything that is not a CPU tensor, or a non-GPU target, takes the ordinary path."""
    if isinstance(t, torch.Tensor) and not t.is_cuda and str(device).startswith("cuda"):
        try:
            return t.pin_memory().to(device, non_blocking=True)
        except RuntimeError:          # no pinned memory to be had: the blocking copy
            pass
    return t.to(device)


def _prefix_lengths(enc, tokenizer):
    """int32 [B] token counts of a tokenizer batch (still on the host) when its mask is a prefix mask — right padding,
    the tokenizers' default — else None.  Handed to the written-out forwards so that they need not inspect the mask on
    the device (a host sync per forward)."""
    if getattr(tokenizer, "padding_side", "right") != "right" or "attention_mask" not in enc:
        return None
    m = enc["attention_mask"]
    if not isinstance(m, torch.Tensor) or m.is_cuda or m.dim() != 2:
        return None
    lens = m.sum(1)
    if not bool((m.bool() == (torch.arange(m.shape[1])[None, :] < lens[:, None])).all()):
        return None
    return lens.to(torch.int32)


def _lean_for_graph(model, amp_dtype, classifier: bool):
    """The written-out forward to capture instead of the transformers module (16-bit AMP, supported family), or None."""
    if amp_dtype not in (torch.bfloat16, torch.float16):
        return None
    if not classifier:
        return lean_encoder_for(model, amp_dtype) or None
    cache = model.__dict__.setdefault("_ts_lean_classifiers", {})
    if amp_dtype not in cache:
        try:
            cache[amp_dtype] = LeanBertClassifier(model, amp_dtype)
        except Exception:
            cache[amp_dtype] = False
    return cache[amp_dtype] or None


class GraphedForward:
    """Small encoder forwards (one query, or a batch of up to 128 queries) replayed from captured HIP graphs.

    A single query through a 12-22 layer encoder is ~200 tiny kernels: launch-bound
    (4.6 ms eager vs 2.5 ms replayed for BERT-base, 8.1 vs 5.0 ms for ModernBERT-base on
    MI355X with the transformers modules, outputs bit-identical in tools/graph_probe.py; the
    written-out forwards: 3.4 ms eager for ModernBERT-base).  The batch is padded to a
    (rows, length) bucket — extra columns are pad ids under attention-mask 0, extra rows a pad
    row with one attended position, so valid positions are unchanged — one graph per bucket is
    captured on first use and replayed afterwards.  The attention mask of the batch must be a
    right-padding mask (what tokenizers produce).  If a model cannot be captured the eager GPU
    forward is used (and remembered)."""

    BUCKETS = (8, 16, 32, 64, 128, 192, 256, 384, 512)
    ROWS = (1, 8, 16, 32, 64, 128)
    MAX_BATCH_TOKENS = 16384          # rows x length bucket of a batch graph (query batches are short)

    def __init__(self, model, pad_token_id: int = 0, amp_dtype=None):
        self.model = model
        self.pad = int(pad_token_id or 0)
        self.amp_dtype = amp_dtype
        self.lean_forward = True      # capture the written-out forward (LeanBertEncoder / LeanModernBertEncoder) when there is one
        self._graphs: Dict[Tuple[int, int], Tuple[Any, torch.Tensor, torch.Tensor, torch.Tensor]] = {}
        self._broken = False

    def _run(self, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        # Autocast keeps low-precision copies of the weights in a cache that is freed when
        # the OUTERMOST autocast context exits.  A graph captured while such a copy is
        # cached would keep reading that freed memory on every replay, so the forward runs
        # outside any caller context and with the cache off (the casts become graph nodes).
        with torch.no_grad(), torch.autocast("cuda", enabled=False):
            lean = _lean_for_graph(self.model, self.amp_dtype, False) if self.lean_forward else None
            if lean is not None:    # (a right-padding mask: no look at it on the host)
                return lean(ids, mask, None, lengths=mask.sum(1, dtype=torch.int32))
            if self.amp_dtype is not None:
                with torch.autocast("cuda", dtype=self.amp_dtype, cache_enabled=False):
                    return self.model(input_ids=ids, attention_mask=mask).last_hidden_state
            return self.model(input_ids=ids, attention_mask=mask).last_hidden_state

    def _capture(self, R: int, L: int, device):
        ids = torch.full((R, L), self.pad, dtype=torch.long, device=device)
        mask = torch.zeros((R, L), dtype=torch.long, device=device)
        mask[:, 0] = 1
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._run(ids, mask)
        torch.cuda.current_stream(device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._run(ids, mask)
        return g, ids, mask, out

    def __call__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """input_ids / attention_mask [B, n] on the GPU (right-padded) -> last_hidden_state [B, n, H] (a copy)."""
        B, n = int(input_ids.shape[0]), int(input_ids.shape[1])
        bucket = next((b for b in self.BUCKETS if b >= n), None)
        rows = next((r for r in self.ROWS if r >= B), None)
        if self._broken or bucket is None or rows is None or (rows > 1 and rows * bucket > self.MAX_BATCH_TOKENS):
            return self._run(input_ids, attention_mask)       # (big batches are not launch-bound; their graphs would pin GBs)
        key = (rows, bucket)
        if key not in self._graphs:
            try:
                self._graphs[key] = self._capture(rows, bucket, input_ids.device)
            except Exception:
                self._broken = True
                torch.cuda.synchronize()
                return self._run(input_ids, attention_mask)
        g, ids, mask, out = self._graphs[key]
        ids.fill_(self.pad)
        mask.zero_()
        mask[:, 0] = 1
        ids[:B, :n].copy_(input_ids)
        mask[:B, :n].copy_(attention_mask)
        g.replay()
        return out[:B, :n].clone()


class GraphedClassifier:
    """One query's (query, candidate) pairs through a sequence-classification model as ONE
    replayed HIP graph: the pairs are padded to a (rows, length) bucket — extra rows are
    copies of the pad row with a single attended position, extra columns are masked — so the
    ~100-pair rerank of a query is one launch-free forward instead of several launch-bound
    eager ones.  Falls back to the eager forward (and remembers) if capture fails."""

    ROWS = (16, 32, 64, 128, 256)
    COLS = (32, 64, 128, 192, 256, 384, 512)

    def __init__(self, model, pad_token_id: int = 0, amp_dtype=None, use_token_types: bool = True):
        self.model = model
        self.pad = int(pad_token_id or 0)
        self.amp_dtype = amp_dtype
        self.use_token_types = use_token_types
        self.lean_forward = True
        self._graphs: Dict[Tuple[int, int], Any] = {}
        self._broken = False

    def _run(self, ids, mask, types):
        kw = {"input_ids": ids, "attention_mask": mask}
        if types is not None:
            kw["token_type_ids"] = types
        with torch.no_grad(), torch.autocast("cuda", enabled=False):   # see GraphedForward._run
            lean = _lean_for_graph(self.model, self.amp_dtype, True) if self.lean_forward else None
            if lean is not None:
                return lean(ids, mask, types, lengths=mask.sum(1, dtype=torch.int32)).reshape(ids.shape[0], -1)
            if self.amp_dtype is not None:
                with torch.autocast("cuda", dtype=self.amp_dtype, cache_enabled=False):
                    return self.model(**kw).logits.float()
            return self.model(**kw).logits.float()

    def _capture(self, R: int, L: int, device):
        ids = torch.full((R, L), self.pad, dtype=torch.long, device=device)
        mask = torch.zeros((R, L), dtype=torch.long, device=device)
        mask[:, 0] = 1
        types = torch.zeros((R, L), dtype=torch.long, device=device) if self.use_token_types else None
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._run(ids, mask, types)
        torch.cuda.current_stream(device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._run(ids, mask, types)
        return g, ids, mask, types, out

    def __call__(self, enc: Dict[str, torch.Tensor]) -> Optional[torch.Tensor]:
        """enc: input_ids / attention_mask (/ token_type_ids) [P, n] on the GPU -> logits [P, labels]
        (a copy), or None when the shape has no bucket (caller runs eagerly)."""
        ids_in, mask_in = enc["input_ids"], enc["attention_mask"]
        P, n = int(ids_in.shape[0]), int(ids_in.shape[1])
        R = next((b for b in self.ROWS if b >= P), None)
        L = next((b for b in self.COLS if b >= n), None)
        if self._broken or R is None or L is None:
            return None
        key = (R, L)
        if key not in self._graphs:
            try:
                self._graphs[key] = self._capture(R, L, ids_in.device)
            except Exception:
                self._broken = True
                torch.cuda.synchronize()
                return None
        g, ids, mask, types, out = self._graphs[key]
        ids.fill_(self.pad)
        mask.zero_()
        mask[:, 0] = 1
        ids[:P, :n].copy_(ids_in)
        mask[:P, :n].copy_(mask_in)
        if types is not None:
            types.zero_()
            if "token_type_ids" in enc:
                types[:P, :n].copy_(enc["token_type_ids"])
        g.replay()
        return out[:P].clone()


# --------------------------------------------------------------------------- bi-encoder
class SentenceEncoder:
    """Stand-in for sentence_transformers.SentenceTransformer (see module docstring)."""

    def __init__(self, model_name: str, device: str = "auto", cache_folder: str = "./models",
                 max_seq_length: Optional[int] = None, amp_dtype=torch.bfloat16, seed: int = 0,
                 use_hip_graph: bool = False):
        self.model_name = model_name
        self.device = resolve_device(device)
        self.use_hip_graph = use_hip_graph
        self._graphed: Optional[GraphedForward] = None
        self.tokenizer, self.model, src = load_backbone(model_name, cache_folder, "base", seed=seed)
        self.model.to(self.device).eval()
        self.amp_dtype = amp_dtype
        self.pooling_mode = "mean"
        self.normalize = False
        self.lean_forward = True      # BERT-family models under 16-bit autocast on a GPU: the written-out forward (LeanBertEncoder)
        self.dense: List[torch.nn.Module] = []
        self.max_seq_length = int(max_seq_length or min(getattr(self.tokenizer, "model_max_length", 512), 512))
        if src:
            self._read_st_modules(src)

    # sentence-transformers directory layout: modules.json + N_<Module>/config.json
    def _read_st_modules(self, src: str) -> None:
        mj = os.path.join(src, "modules.json")
        if not os.path.exists(mj):
            return
        for mod in json.load(open(mj)):
            path, typ = os.path.join(src, mod.get("path", "")), mod.get("type", "")
            cfgp = os.path.join(path, "config.json")
            if typ.endswith("Pooling") and os.path.exists(cfgp):
                c = json.load(open(cfgp))
                if c.get("pooling_mode_cls_token"):
                    self.pooling_mode = "cls"
                elif c.get("pooling_mode_max_tokens"):
                    self.pooling_mode = "max"
                else:
                    self.pooling_mode = "mean"
            elif typ.endswith("Dense") and os.path.exists(cfgp):
                c = json.load(open(cfgp))
                lin = torch.nn.Linear(c["in_features"], c["out_features"], bias=c.get("bias", True))
                wp = os.path.join(path, "model.safetensors")
                if os.path.exists(wp):
                    from safetensors.torch import load_file
                    sd = load_file(wp)
                    lin.load_state_dict({k.replace("linear.", ""): v for k, v in sd.items()})
                act = c.get("activation_function", "torch.nn.modules.linear.Identity")
                seq = [lin] + ([torch.nn.Tanh()] if act.endswith("Tanh") else [])
                self.dense.append(torch.nn.Sequential(*seq).to(self.device).eval())
            elif typ.endswith("Normalize"):
                self.normalize = True
            elif typ.endswith("Transformer"):
                sc = os.path.join(path, "sentence_bert_config.json")
                if os.path.exists(sc):
                    self.max_seq_length = int(json.load(open(sc)).get("max_seq_length", self.max_seq_length))

    def get_sentence_embedding_dimension(self) -> int:
        if self.dense:
            return int(self.dense[-1][0].out_features)
        return int(self.model.config.hidden_size)

    def _pool(self, hidden: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        if self.pooling_mode == "cls":
            return hidden[:, 0]
        m = mask.unsqueeze(-1).to(hidden.dtype)
        if self.pooling_mode == "max":
            return hidden.masked_fill(m == 0, -1e9).max(dim=1).values
        return (hidden * m).sum(1) / m.sum(1).clamp(min=1e-9)

    @torch.no_grad()
    def encode(self, sentences: Union[str, Sequence[str]], batch_size: int = 32,
               convert_to_numpy: bool = True, convert_to_tensor: bool = False,
               show_progress_bar: bool = False, normalize_embeddings: bool = False, **_):
        single = isinstance(sentences, str)
        texts = [sentences] if single else list(sentences)
        order = np.argsort([-len(t) for t in texts], kind="stable")  # length-sorted batches
        res = torch.zeros((len(texts), self.get_sentence_embedding_dimension()), dtype=torch.float32,
                          device=self.device)
        for s in range(0, len(texts), batch_size):
            idx = order[s:s + batch_size]
            enc = self.tokenizer([texts[i] for i in idx], truncation=True, padding=True,
                                 max_length=self.max_seq_length, return_tensors="pt")
            lengths = _prefix_lengths(enc, self.tokenizer)
            enc = {k: to_device_async(v, self.device) for k, v in enc.items() if k in ("input_ids", "attention_mask", "token_type_ids")}
            if "token_type_ids" in enc and not hasattr(self.model.config, "type_vocab_size"):
                enc.pop("token_type_ids")
            if (self.use_hip_graph and str(self.device).startswith("cuda") and
                    (len(idx) == 1 or (len(idx) <= GraphedForward.ROWS[-1] and lengths is not None))):
                if self._graphed is None:  # the autocast state of the first call is baked into the graphs
                    self._graphed = GraphedForward(
                        self.model, getattr(self.tokenizer, "pad_token_id", 0),
                        torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None)
                hidden = self._graphed(enc["input_ids"], enc["attention_mask"])
            else:
                lean = False
                if self.lean_forward and str(self.device).startswith("cuda") and torch.is_autocast_enabled("cuda"):
                    lean = lean_encoder_for(self.model, torch.get_autocast_dtype("cuda"))
                hidden = (lean(enc["input_ids"], enc["attention_mask"], enc.get("token_type_ids"),
                               lengths=to_device_async(lengths, self.device) if lengths is not None else None) if lean
                          else self.model(**enc).last_hidden_state)
            emb = self._pool(hidden.float(), enc["attention_mask"])
            for d in self.dense:
                emb = d(emb)
            if self.normalize or normalize_embeddings:
                emb = F.normalize(emb, p=2, dim=1)
            # back into the caller's order: one scatter per batch (not one tiny copy per text)
            res[to_device_async(torch.from_numpy(np.ascontiguousarray(idx)), res.device)] = emb.float()
        if single:
            res = res[0]
        if convert_to_tensor:
            return res
        return res.cpu().numpy() if convert_to_numpy else list(res)


# --------------------------------------------------------------------------- pair assembly on the device
class PairAssembler:
    """(query, document) cross-encoder inputs built from TOKEN IDS on the GPU.

    The reference hands text pairs to the tokenizer for every query (reference
    src/stage3_reranker.py:113-118, 139-160): 100 documents are re-tokenised per query on the host.
    Here every document is tokenised ONCE (at add_documents), its ids are kept in a padded int32
    table in HBM, a query is tokenised once, and the batch tensors ``[prefix] q [middle] d [suffix]``
    are gathered on the device, with the tokenizer's own longest-first truncation restated in
    closed form.

    Nothing about the tokenizer is assumed: the special-token template, the token types and the
    truncation rule are PROBED from it (``tokenizer(a, b, truncation=True, max_length=...)`` on
    synthetic probes, including over-long and equal-length cases) and the assembler only declares
    itself usable (``self.ok``) if it reproduces the tokenizer's output on every probe; otherwise the
    caller keeps tokenising text pairs."""

    RULES = ("second_on_ties", "longer_gets_odd")   # transformers' python loop / the tokenizers crate's closed form

    def __init__(self, tokenizer, max_length: int, pad_token_id: Optional[int] = None):
        self.tok = tokenizer
        self.max_length = int(max_length)
        self.pad = int(pad_token_id if pad_token_id is not None else (getattr(tokenizer, "pad_token_id", 0) or 0))
        self.prefix: List[int] = []
        self.middle: List[int] = []
        self.suffix: List[int] = []
        self.type_b = 0
        self.has_types = False
        self.rule: Optional[str] = None
        self.ok = False
        self.why = ""
        self._doc_ids: List[List[int]] = []     # per document slot: the first max_length ids ...
        self._doc_len: List[int] = []           # ... and the true token count
        self._table = None                      # (device, int32 [N, L] padded ids, int32 [N] lengths)
        try:
            self._probe()
        except Exception as e:  # a tokenizer we cannot characterise: the caller keeps the text path
            self.ok, self.why = False, f"probe failed: {e!r}"

    # -- the tokenizer, one text at a time -----------------------------------------
    def ids_of(self, text: str) -> List[int]:
        """Token ids of one text WITHOUT special tokens and without truncation (the truncation rule needs the
        true lengths of both texts; only the first max_length ids can ever be used)."""
        if isinstance(self.tok, HashTokenizer):
            return self.tok._ids(text)
        return list(self.tok(text, add_special_tokens=False, truncation=False)["input_ids"])

    def _pair(self, a: str, b: str) -> Dict[str, List[int]]:
        enc = self.tok(a, b, truncation=True, max_length=self.max_length, padding=False)
        out = {}
        for k, v in enc.items():
            if k in ("input_ids", "token_type_ids"):
                v = list(v)
                out[k] = list(v[0]) if v and isinstance(v[0], (list, tuple)) else v   # (HashTokenizer returns rows)
        return out

    @staticmethod
    def _find(hay: List[int], needle: List[int], start: int = 0) -> int:
        n = len(needle)
        for i in range(start, len(hay) - n + 1):
            if hay[i:i + n] == needle:
                return i
        return -1

    def _probe(self) -> None:
        words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel"]
        a_txt, b_txt = "alpha bravo charlie", "delta echo foxtrot golf"
        a, b = self.ids_of(a_txt), self.ids_of(b_txt)
        if not a or not b:
            self.why = "empty probe tokenisation"
            return
        enc = self._pair(a_txt, b_txt)
        ids = enc["input_ids"]
        ia = self._find(ids, a)
        ib = self._find(ids, b, ia + len(a)) if ia >= 0 else -1
        if ia < 0 or ib < 0:
            self.why = "a text tokenises differently inside a pair"
            return
        self.prefix, self.middle, self.suffix = ids[:ia], ids[ia + len(a): ib], ids[ib + len(b):]
        tt = enc.get("token_type_ids")
        self.has_types = tt is not None
        if tt is not None:
            if any(tt[:ib]) or len(set(tt[ib:])) != 1:
                self.why = "unexpected token-type layout"
                return
            self.type_b = int(tt[ib])
        # truncation rule: probes that overflow max_length in every regime
        budget = self.max_length - len(self.prefix) - len(self.middle) - len(self.suffix)
        if budget < 2:
            self.why = "max_length leaves no room"
            return

        def text_of(n, off):   # n words cycling through the list: ids_of() gives >= n tokens
            return " ".join(words[(off + i) % len(words)] for i in range(n))
        probes = [(3, budget + 5), (budget + 5, 3), (budget, budget), (budget // 2 + 1, budget // 2 + 1),
                  (budget // 2 + 2, budget // 2 + 1), (budget // 2 + 1, budget // 2 + 2), (budget - 1, 2), (2, budget - 1),
                  (budget // 2 + 7, budget // 2 + 2), (budget + 9, budget + 4), (1, 1), (budget // 3, budget)]
        cases = []
        for na, nb in probes:
            ta, tb = text_of(max(na, 1), 0), text_of(max(nb, 1), 3)
            cases.append((self.ids_of(ta), self.ids_of(tb), self._pair(ta, tb)))
        for rule in self.RULES:
            good = True
            for qa, qb, want in cases:
                la, lb = self.truncated_lengths(len(qa), [len(qb)], rule)
                got = self.prefix + qa[:la] + self.middle + qb[: lb[0]] + self.suffix
                types = [0] * (len(self.prefix) + la + len(self.middle)) + [self.type_b] * (lb[0] + len(self.suffix))
                if got != want["input_ids"] or (self.has_types and types != want["token_type_ids"]):
                    good = False
                    break
            if good:
                self.rule, self.ok = rule, True
                return
        self.why = "no known truncation rule reproduces the tokenizer"

    # -- longest-first truncation in closed form -----------------------------------
    def truncated_lengths(self, la: int, lb, rule: Optional[str] = None):
        """Lengths kept of a query of `la` tokens paired with documents of `lb` tokens (sequence or
        tensor) under longest-first truncation to max_length.  Returns (la' per document, lb')."""
        rule = rule or self.rule
        budget = self.max_length - len(self.prefix) - len(self.middle) - len(self.suffix)
        is_t = torch.is_tensor(lb)
        lbv = lb.to(torch.int64) if is_t else torch.as_tensor(list(lb), dtype=torch.int64)
        lav = torch.full_like(lbv, int(la))
        lav, lbv = self._longest_first(lav, lbv, budget, rule)
        if is_t:
            return lav, lbv
        if len(lbv) == 1:
            return int(lav[0]), [int(lbv[0])]
        return [int(x) for x in lav], [int(x) for x in lbv]

    @staticmethod
    def _longest_first(la, lb, budget: int, rule: str):
        """int64 tensors la, lb -> kept lengths.  Both known implementations first shorten the LONGER text
        until the two are equal (or the excess is gone); they differ in how an odd remainder is split:
          second_on_ties    transformers' python loop: one token at a time, from the second text on ties
                            -> the second text loses ceil(e/2), the first floor(e/2);
          longer_gets_odd   the tokenizers crate's closed form: floor(budget/2) for the originally shorter
                            text (the FIRST on a tie), ceil(budget/2) for the originally longer one."""
        first_longer = la > lb
        e = (la + lb - budget).clamp(min=0)
        c = torch.minimum(e, (la - lb).clamp(min=0))
        la, e = la - c, e - c
        c = torch.minimum(e, (lb - la).clamp(min=0))
        lb, e = lb - c, e - c
        if rule == "second_on_ties":
            lb, la = lb - (e + 1) // 2, la - e // 2
        else:
            big, small = (e + 1) // 2, e // 2                     # what the two sides lose
            la = la - torch.where(first_longer, small, big)
            lb = lb - torch.where(first_longer, big, small)
        return la.clamp(min=0), lb.clamp(min=0)

    # -- document table ---------------------------------------------------------------
    def add_documents(self, texts: Sequence[str]) -> None:
        for t in texts:
            ids = self.ids_of(t)
            self._doc_len.append(len(ids))
            self._doc_ids.append(ids[: self.max_length])
        self._table = None

    def __len__(self) -> int:
        return len(self._doc_ids)

    def table(self, device):
        if self._table is None or self._table[0] != str(device):
            n = len(self._doc_ids)
            L = max(1, max((len(x) for x in self._doc_ids), default=1))
            ids = np.full((n, L), self.pad, dtype=np.int32)
            lens = np.zeros((n,), dtype=np.int32)
            for i, x in enumerate(self._doc_ids):
                ids[i, : len(x)] = x
            lens[:] = self._doc_len
            self._table = (str(device), torch.from_numpy(ids).to(device), torch.from_numpy(lens).to(device))
        return self._table[1], self._table[2]

    def _device_consts(self, device):
        """The template's special-token runs as device tensors, uploaded once per device: a host-to-device copy inside
        batch() would be a host sync per cross-encoder forward (the GPU then idles while the next forward is enqueued)."""
        cache = self.__dict__.setdefault("_consts", {})
        key = str(device)
        if key not in cache:
            cache[key] = {tuple(v): torch.as_tensor(list(v), dtype=torch.int64, device=device)
                          for v in (self.prefix, self.middle, self.suffix) if v}
            cache[key][()] = torch.zeros((1, 1), dtype=torch.int64, device=device)
        return cache[key]

    # -- batch tensors -----------------------------------------------------------------
    def plan(self, q_ids: List[List[int]], pair_q, pair_slot, device):
        """Everything that does not depend on the batch cut: per-pair kept lengths and total length.
        pair_q / pair_slot: int64 tensors [P] (query number, document slot)."""
        dtab, dlen = self.table(device)
        Lq = max(1, min(self.max_length, max(len(x) for x in q_ids)))
        qtab = torch.full((len(q_ids), Lq), self.pad, dtype=torch.int32)
        for i, x in enumerate(q_ids):
            x = x[: self.max_length]
            qtab[i, : len(x)] = torch.as_tensor(x, dtype=torch.int32)
        # (through pinned memory: a pageable host-to-device copy waits for everything already queued on the stream —
        # here the whole of stages 1 and 2 — before the host may go on planning stage 3)
        qtab = to_device_async(qtab, device)
        qlen = to_device_async(torch.as_tensor([len(x) for x in q_ids], dtype=torch.int64), device)
        budget = self.max_length - len(self.prefix) - len(self.middle) - len(self.suffix)
        la, lb = self._longest_first(qlen[pair_q], dlen[pair_slot].to(torch.int64), budget, self.rule)
        total = la + lb + (len(self.prefix) + len(self.middle) + len(self.suffix))
        return {"qtab": qtab, "dtab": dtab, "la": la, "lb": lb, "total": total, "pair_q": pair_q, "pair_slot": pair_slot}

    def batch(self, plan, sel, width: Optional[int] = None):
        """input_ids / attention_mask / token_type_ids [len(sel), L] for the pairs `sel` (int64 tensor)."""
        dev = plan["la"].device
        la, lb = plan["la"][sel][:, None], plan["lb"][sel][:, None]
        L = int(width) if width is not None else int(plan["total"][sel].max().item())
        t = torch.arange(L, device=dev)[None, :]
        npre, nmid, nsuf = len(self.prefix), len(self.middle), len(self.suffix)
        a0 = npre
        a1 = a0 + la
        a2 = a1 + nmid
        a3 = a2 + lb
        a4 = a3 + nsuf
        qrows = plan["qtab"][plan["pair_q"][sel]].to(torch.int64)
        drows = plan["dtab"][plan["pair_slot"][sel]].to(torch.int64)
        qpart = torch.gather(qrows, 1, (t - a0).clamp(0, qrows.shape[1] - 1).expand(len(sel), L))
        dpart = torch.gather(drows, 1, (t - a2).clamp(0, drows.shape[1] - 1).expand(len(sel), L))

        consts = self._device_consts(dev)

        def const(vals, off):   # special tokens at positions off + i
            if not vals:
                return consts[()]
            return consts[tuple(vals)][(t - off).clamp(0, len(vals) - 1)]
        ids = torch.full((len(sel), L), self.pad, dtype=torch.int64, device=dev)
        ids = torch.where(t < a4, const(self.suffix, a3), ids)
        ids = torch.where(t < a3, dpart, ids)
        ids = torch.where(t < a2, const(self.middle, a1), ids)
        ids = torch.where(t < a1, qpart, ids)
        ids = torch.where(t < a0, const(self.prefix, 0), ids)
        mask = (t < a4).to(torch.int64)
        # "lengths": the mask is a prefix mask by construction — the written-out forward needs no look at it (no host sync)
        out = {"input_ids": ids, "attention_mask": mask, "lengths": a4.reshape(-1).clamp(max=L).to(torch.int32)}
        if self.has_types:
            out["token_type_ids"] = ((t >= a2) & (t < a4)).to(torch.int64) * self.type_b
        return out


    def batch_packed(self, plan, sel, n_tokens: int, width: int):
        """The pairs `sel` as ONE packed token sequence (what LeanBertEncoder.hidden_packed takes): input_ids / positions
        (index inside the pair) / token_type_ids int64 [n_tokens], lengths / offsets int32 [len(sel)].  ``n_tokens`` (the
        sum of the pairs' lengths) and ``width`` (their maximum) come from the host so that nothing here synchronises."""
        dev = plan["la"].device
        n = int(sel.shape[0])
        la, lb = plan["la"][sel], plan["lb"][sel]
        npre, nmid, nsuf = len(self.prefix), len(self.middle), len(self.suffix)
        lens = la + lb + (npre + nmid + nsuf)
        offs = torch.cumsum(lens, 0) - lens
        seq = torch.repeat_interleave(torch.arange(n, device=dev), lens, output_size=int(n_tokens))
        t = torch.arange(int(n_tokens), device=dev) - offs[seq]
        a0 = npre
        a1 = a0 + la[seq]
        a2 = a1 + nmid
        a3 = a2 + lb[seq]
        qtab, dtab = plan["qtab"], plan["dtab"]
        qpart = qtab[plan["pair_q"][sel][seq], (t - a0).clamp(0, qtab.shape[1] - 1)].to(torch.int64)
        dpart = dtab[plan["pair_slot"][sel][seq], (t - a2).clamp(0, dtab.shape[1] - 1)].to(torch.int64)
        consts = self._device_consts(dev)

        def const(vals, off):
            if not vals:
                return consts[()].reshape(())
            return consts[tuple(vals)][(t - off).clamp(0, len(vals) - 1)]
        ids = const(self.suffix, a3)
        ids = torch.where(t < a3, dpart, ids)
        ids = torch.where(t < a2, const(self.middle, a1), ids)
        ids = torch.where(t < a1, qpart, ids)
        ids = torch.where(t < a0, const(self.prefix, 0), ids)
        out = {"packed": True, "input_ids": ids.contiguous(), "positions": t, "lengths": lens.to(torch.int32),
               "offsets": offs.to(torch.int32), "max_len": int(width)}
        out["token_type_ids"] = ((t >= a2).to(torch.int64) * self.type_b) if self.has_types else None
        return out


# --------------------------------------------------------------------------- lean BERT-family forwards
class LeanBertEncoder:
    """The forward of a BERT / RoBERTa / XLM-R encoder (``AutoModel``: last hidden state) written out in plain torch ops
    on the checkpoint's own weights: the arithmetic ``torch.autocast`` performs on the transformers module
    (linears in the compute dtype with outputs in that dtype, residual adds and LayerNorms in fp32, softmax
    inside scaled_dot_product_attention), minus what the module spends around it at 10^5 tokens per batch:
    ONE fused QKV GEMM instead of three (the fp32 LayerNorm output is cast once, not three times), weights
    cast once at construction instead of through the autocast cache, no mask tensor at all for a batch without
    padding, no per-layer Python of the generic module.  ``compute_dtype=None`` runs everything in fp32 and
    reproduces the module's fp32 forward (tests).  With 16-bit compute on a GPU the passes between the GEMMs are
    the library's HIP kernels: embedding gather + LayerNorm, residual add + LayerNorm + cast, and attention over
    each sequence's own tokens (ts_embed_layernorm, ts_add_layernorm, ts_attention_varlen).  Used for the batched
    forwards of all three stages when the model is of this family (SentenceEncoder.encode, ColBERTScorer._forward,
    CrossEncoderModel through LeanBertClassifier); HIP-graph replays and other architectures keep the transformers
    forward."""

    def __init__(self, base, compute_dtype=None):
        cfg = base.config
        self.kind = cfg.model_type
        if self.kind not in ("bert", "roberta", "xlm-roberta"):
            raise ValueError(f"no lean forward for model type {self.kind!r}")
        if getattr(cfg, "position_embedding_type", "absolute") != "absolute" or cfg.hidden_act not in ("gelu", "gelu_new", "relu"):
            raise ValueError("unsupported BERT variant")
        if not hasattr(base, "embeddings") or not hasattr(base, "encoder"):
            raise ValueError("not a bare encoder module")
        self.cd = compute_dtype
        cd = compute_dtype or torch.float32
        self.fused_layernorm = True   # 16-bit compute on a GPU: residual add + LayerNorm + cast as one HIP pass (ts_add_layernorm)
        self.fused_attention = True   # ... and attention over each sequence's own tokens, no mask tensor (ts_attention_varlen)
        emb = base.embeddings
        self.word, self.pos, self.typ = emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight
        self.emb_ln = (emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps)
        self.pad_idx = int(getattr(emb, "padding_idx", 0) or 0)
        self.heads = int(cfg.num_attention_heads)
        self.act = {"gelu": F.gelu, "gelu_new": lambda x: F.gelu(x, approximate="tanh"), "relu": F.relu}[cfg.hidden_act]
        self.layers = []
        for l in base.encoder.layer:
            a = l.attention.self
            self.layers.append({
                "wqkv": torch.cat([a.query.weight, a.key.weight, a.value.weight], 0).detach().to(cd).contiguous(),
                "bqkv": torch.cat([a.query.bias, a.key.bias, a.value.bias], 0).detach().to(cd).contiguous(),
                "wo": l.attention.output.dense.weight.detach().to(cd), "bo": l.attention.output.dense.bias.detach().to(cd),
                "ln1": (l.attention.output.LayerNorm.weight, l.attention.output.LayerNorm.bias, l.attention.output.LayerNorm.eps),
                "w1": l.intermediate.dense.weight.detach().to(cd), "b1": l.intermediate.dense.bias.detach().to(cd),
                "w2": l.output.dense.weight.detach().to(cd), "b2": l.output.dense.bias.detach().to(cd),
                "ln2": (l.output.LayerNorm.weight, l.output.LayerNorm.bias, l.output.LayerNorm.eps)})
        # projections with a short reduction dimension (Q/K/V, attention output, feed-forward up): weights re-tiled once for
        # the streamed-weight kernel (ts_linear_act) when they live on a GPU in a 16-bit type
        self.fused_linear = True
        self.min_linear_rows = 4096   # below this many tokens the forward is launch-bound and the library GEMM is as good
        # ... and BertSelfOutput / BertOutput (projection + residual + LayerNorm) as ONE kernel when a workgroup can own
        # whole rows of the output (H <= 384: ts_linear_add_layernorm; the projection's output never goes to HBM)
        self.fused_output_layernorm = True
        self.gelu_in_down = True          # ... and the erf GELU between the two feed-forward projections inside the second
        self.fused_mlp = True             # ... or, at hidden size 384, the whole feed-forward block as ONE kernel (ts_mlp_add_layernorm)
        for p in self.layers:
            p["tqkv"] = p["to"] = p["t1"] = p["to_ln"] = p["t2_ln"] = None
            if compute_dtype in (torch.bfloat16, torch.float16) and p["wqkv"].is_cuda:
                try:
                    from .index import TiledLinear
                    for key, w, b in (("tqkv", "wqkv", "bqkv"), ("to", "wo", "bo"), ("t1", "w1", "b1")):
                        if TiledLinear.usable(int(p[w].shape[0]), int(p[w].shape[1])):
                            p[key] = TiledLinear(p[w], p[b])
                    for key, w, b in (("to_ln", "wo", "bo"), ("t2_ln", "w2", "b2")):
                        if TiledLinear.usable_with_layernorm(int(p[w].shape[0]), int(p[w].shape[1])):
                            p[key] = p["to"] if key == "to_ln" and p["to"] is not None else TiledLinear(p[w], p[b], with_layernorm=True)
                except Exception:
                    p["tqkv"] = p["to"] = p["t1"] = p["to_ln"] = p["t2_ln"] = None    # (no library: the GEMMs stay with torch)
    @torch.no_grad()
    def hidden(self, input_ids, attention_mask, token_type_ids=None, lengths=None):
        """-> (last hidden state float32 [B, L, H], its copy in the compute dtype).  ``lengths`` (int32 [B], optional):
        the caller's promise that attention_mask[b] is 1 exactly on the first lengths[b] positions.  Rows at padded
        positions hold finite values of no meaning (as in the module's output)."""
        cd = self.cd or torch.float32
        B, L = input_ids.shape
        if self.kind == "bert":
            pos = torch.arange(L, device=input_ids.device)[None, :].expand(B, L)
        else:   # RoBERTa family: positions count the non-padding tokens, offset by the padding index
            nonpad = (input_ids != self.pad_idx).to(torch.int64)
            pos = torch.cumsum(nonpad, dim=1) * nonpad + self.pad_idx
        if L + (0 if self.kind == "bert" else self.pad_idx + 1) > int(self.pos.shape[0]):
            raise ValueError(f"sequence length {L} exceeds the model's {int(self.pos.shape[0])} position embeddings")
        H, nh = int(self.word.shape[-1]), self.heads
        fused = (self.fused_layernorm and self.cd in (torch.bfloat16, torch.float16) and input_ids.is_cuda and H % 4 == 0 and H <= 2048)
        if fused:
            from .index import add_layernorm   # HIP kernel (raises without the library)

            def add_ln(new, old, ln):          # -> (fp32 stream, 16-bit copy for the next GEMM) in one pass
                return add_layernorm(new, old, ln[0], ln[1], ln[2], lp_dtype=cd)
        else:
            def add_ln(new, old, ln):          # the same arithmetic in torch ops (16-bit + fp32 -> fp32 add, fp32 LayerNorm, cast)
                y = F.layer_norm(new + old if old is not None else new, (H,), ln[0], ln[1], ln[2])
                return y, y.to(cd)
        if fused and all(t.dtype == torch.float32 and t.is_contiguous() for t in (self.word, self.pos, self.typ)):
            from .index import embed_layernorm   # gather + sum + LayerNorm + cast in one pass
            x, xb = embed_layernorm(input_ids, pos, token_type_ids, self.word, self.pos, self.typ, *self.emb_ln, lp_dtype=cd)
        else:
            x = self.word[input_ids] + self.typ[token_type_ids if token_type_ids is not None else torch.zeros_like(input_ids)]
            x, xb = add_ln((x + self.pos[pos]).float(), None, self.emb_ln)
        mask = lens = abuf = None
        dh = H // nh
        if (fused and self.fused_attention and dh in (32, 64) and B <= 65535 and L <= (1120 if dh == 32 else 576)):
            # right-padded batch (what tokenizers and PairAssembler produce): every sequence attends over its own tokens
            lens = (lengths.to(torch.int32) if lengths is not None
                    else torch.full((B,), L, dtype=torch.int32, device=x.device) if attention_mask is None
                    else attention_mask.sum(1, dtype=torch.int32))
            if lengths is None and attention_mask is not None and not bool(
                    (attention_mask.to(torch.bool) == (torch.arange(L, device=x.device)[None, :] < lens[:, None])).all()):
                lens = None                     # holes or left padding: the masked torch kernel below
        if lens is not None:
            from .index import attention_varlen
            abuf = torch.zeros((B, L, H), dtype=cd, device=x.device)   # padded rows stay zero through all layers
        elif attention_mask is not None and not bool(attention_mask.all()):
            mask = attention_mask.to(torch.bool)[:, None, None, :]
        tl = fused and self.fused_linear and B * L >= self.min_linear_rows      # streamed-weight projections (ts_linear_act)
        for p in self.layers:
            qkv = p["tqkv"](xb) if tl and p["tqkv"] is not None else F.linear(xb, p["wqkv"], p["bqkv"])
            if lens is not None:
                a = attention_varlen(qkv, lens, nh, out=abuf)
            else:
                qkv = qkv.view(B, L, 3, nh, dh)
                q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))          # [B, heads, L, dh] views
                a = F.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(B, L, H)
            fo = tl and self.fused_output_layernorm
            if fo and p["to_ln"] is not None and a.is_contiguous():
                x, xb = p["to_ln"].add_layernorm(a, x, *p["ln1"])
            else:
                o = p["to"](a) if tl and p["to"] is not None and a.is_contiguous() else F.linear(a, p["wo"], p["bo"])
                x, xb = add_ln(o, x, p["ln1"])
            if fo and p["t2_ln"] is not None:
                x, xb = self._down_ln(p, xb, x, tl)
            else:
                f = F.linear(self._up(p, xb, tl), p["w2"], p["b2"])
                x, xb = add_ln(f, x, p["ln2"])
        return x, xb

    def _down_ln(self, p, xb, x, tl: bool):
        """BertIntermediate + BertOutput: up projection, activation, down projection, residual add, LayerNorm — two
        kernels when the activation is the erf GELU: the up projection writes its output BEFORE the activation and the
        down kernel applies the GELU while it stages its rows (the erf is vector-ALU work: in the up projection's epilogue
        it is exposed, beside the down projection's matrix instructions it is not).  Same bits either way."""
        if tl and p["t1"] is not None and self.act is F.gelu:
            if self.fused_mlp:
                from .index import mlp_add_layernorm, mlp_usable
                if mlp_usable(p["t1"], p["t2_ln"]):
                    return mlp_add_layernorm(p["t1"], p["t2_ln"], xb, x, *p["ln2"])    # the whole block: one kernel
            if self.gelu_in_down:
                return p["t2_ln"].add_layernorm(p["t1"](xb), x, *p["ln2"], gelu_input=True)
        return p["t2_ln"].add_layernorm(self._up(p, xb, tl), x, *p["ln2"])

    def _up(self, p, xb, tl: bool):
        """The feed-forward up projection with its activation: one kernel (GELU in the epilogue) when it applies."""
        if tl and p["t1"] is not None:
            return p["t1"](xb, gelu=True) if self.act is F.gelu else self.act(p["t1"](xb))
        return self.act(F.linear(xb, p["w1"], p["b1"]))

    def __call__(self, input_ids, attention_mask, token_type_ids=None, lengths=None) -> torch.Tensor:
        return self.hidden(input_ids, attention_mask, token_type_ids, lengths)[0]

    # -- packed batches: the sequences' tokens concatenated, no padded position anywhere --------------------------
    def packed_ok(self, device, max_len: int, n_seq: int) -> bool:
        """Whether hidden_packed() can run: 16-bit compute on a GPU with every HIP kernel applicable."""
        H, nh = int(self.word.shape[-1]), self.heads
        dh = H // nh
        return (self.fused_layernorm and self.fused_attention and self.cd in (torch.bfloat16, torch.float16) and
                str(device).startswith("cuda") and H % 4 == 0 and H <= 2048 and dh in (32, 64) and n_seq <= 65535 and
                max_len <= (1120 if dh == 32 else 576) and
                max_len + (0 if self.kind == "bert" else self.pad_idx + 1) <= int(self.pos.shape[0]) and
                all(t.dtype == torch.float32 and t.is_contiguous() for t in (self.word, self.pos, self.typ)))

    @torch.no_grad()
    def hidden_packed(self, input_ids, positions, token_type_ids, lengths, offsets, max_len: int):
        """The same forward on a PACKED batch: input_ids / positions (index of the token inside its sequence) /
        token_type_ids int64 [T] — all sequences' tokens one after the other —, lengths / offsets int32 [B] (token count
        and first token of each sequence).  GEMMs, LayerNorms and GELU see T = sum(lengths) rows instead of
        B x max(lengths) (a length-sorted batch of 1024 reranking pairs: 10 % fewer), attention works per sequence as
        before.  -> (last hidden state float32 [T, H], its 16-bit copy).  packed_ok() must hold."""
        from .index import add_layernorm, attention_varlen, embed_layernorm
        cd, nh = self.cd, self.heads
        pos = positions if self.kind == "bert" else positions + (self.pad_idx + 1)
        x, xb = embed_layernorm(input_ids, pos, token_type_ids, self.word, self.pos, self.typ, *self.emb_ln, lp_dtype=cd)
        abuf = torch.empty((int(input_ids.shape[0]), int(self.word.shape[-1])), dtype=cd, device=input_ids.device)
        tl = self.fused_linear and int(input_ids.shape[0]) >= self.min_linear_rows
        for p in self.layers:
            qkv = p["tqkv"](xb) if tl and p["tqkv"] is not None else F.linear(xb, p["wqkv"], p["bqkv"])
            a = attention_varlen(qkv, lengths, nh, out=abuf, offs=offsets, max_len=max_len)   # (every row is a valid token)
            fo = tl and self.fused_output_layernorm
            if fo and p["to_ln"] is not None:
                x, xb = p["to_ln"].add_layernorm(a, x, *p["ln1"])
            else:
                o = p["to"](a) if tl and p["to"] is not None else F.linear(a, p["wo"], p["bo"])
                x, xb = add_layernorm(o, x, *p["ln1"], lp_dtype=cd)
            if fo and p["t2_ln"] is not None:
                x, xb = self._down_ln(p, xb, x, tl)
            else:
                f = F.linear(self._up(p, xb, tl), p["w2"], p["b2"])
                x, xb = add_layernorm(f, x, *p["ln2"], lp_dtype=cd)
        return x, xb


class LeanBertClassifier(LeanBertEncoder):
    """LeanBertEncoder + the pooler / classification head of a ...ForSequenceClassification checkpoint -> logits."""

    def __init__(self, hf_model, compute_dtype=None):
        kind = hf_model.config.model_type
        if kind not in ("bert", "roberta", "xlm-roberta"):
            raise ValueError(f"no lean forward for model type {kind!r}")
        base = hf_model.bert if kind == "bert" else hf_model.roberta
        super().__init__(base, compute_dtype)
        cd = compute_dtype or torch.float32
        if kind == "bert":
            self.head = [(base.pooler.dense.weight.detach().to(cd), base.pooler.dense.bias.detach().to(cd), True),
                         (hf_model.classifier.weight.detach().to(cd), hf_model.classifier.bias.detach().to(cd), False)]
        else:
            c = hf_model.classifier
            self.head = [(c.dense.weight.detach().to(cd), c.dense.bias.detach().to(cd), True),
                         (c.out_proj.weight.detach().to(cd), c.out_proj.bias.detach().to(cd), False)]

    @torch.no_grad()
    def __call__(self, input_ids, attention_mask, token_type_ids=None, lengths=None) -> torch.Tensor:
        y = self.hidden(input_ids, attention_mask, token_type_ids, lengths)[1][:, 0]
        for w, b, tanh in self.head:
            y = F.linear(y, w, b)
            if tanh:
                y = torch.tanh(y)
        return y.float()

    @torch.no_grad()
    def logits_packed(self, input_ids, positions, token_type_ids, lengths, offsets, max_len: int) -> torch.Tensor:
        y = self.hidden_packed(input_ids, positions, token_type_ids, lengths, offsets, max_len)[1][offsets.long()]   # [CLS] rows
        for w, b, tanh in self.head:
            y = F.linear(y, w, b)
            if tanh:
                y = torch.tanh(y)
        return y.float()


class LeanModernBertEncoder:
    """The forward of a ModernBERT encoder (the reference's default stage-2 token encoder, GTE-ModernColBERT:
    src/stage2_rescorer.py:30) written out the same way: pre-LN blocks, fused Wqkv, rotary embedding computed in fp32
    from the module's own inverse frequencies, global layers and local layers with the bidirectional window
    ``|q - k| <= local_attention / 2``, gated-GELU MLP, final LayerNorm — the arithmetic transformers performs under
    ``torch.autocast`` (linears and attention in the compute dtype, LayerNorms and the residual stream in fp32).  The
    module launches ~40 small kernels per layer; here a layer is four GEMMs and four HIP kernels (ts_attention_varlen
    with the rotary tables and its window, ts_add_prenorm twice, ts_geglu).  ``compute_dtype=None``: fp32 torch ops
    (tests)."""

    def __init__(self, base, compute_dtype=None):
        cfg = base.config
        if cfg.model_type != "modernbert" or not hasattr(base, "layers") or not hasattr(base, "rotary_emb"):
            raise ValueError("not a ModernBERT encoder module")
        if getattr(cfg, "hidden_activation", "gelu") != "gelu":
            raise ValueError("unsupported ModernBERT activation")
        self.kind = "modernbert"
        self.cd = compute_dtype
        cd = compute_dtype or torch.float32
        self.fused = True             # 16-bit compute on a GPU: the HIP kernels named above
        self.heads = int(cfg.num_attention_heads)
        self.window = int(cfg.sliding_window)
        self.rotary = base.rotary_emb
        emb = base.embeddings
        self.tok = emb.tok_embeddings.weight
        ln = lambda m: None if isinstance(m, torch.nn.Identity) else (m.weight, m.bias, m.eps)
        lin = lambda m: (m.weight.detach().to(cd).contiguous(), m.bias.detach().to(cd) if m.bias is not None else None)
        self.emb_ln = ln(emb.norm)
        self.final_ln = ln(base.final_norm)
        self.layers = [{"ln1": ln(l.attn_norm), "qkv": lin(l.attn.Wqkv), "o": lin(l.attn.Wo), "ln2": ln(l.mlp_norm),
                        "wi": lin(l.mlp.Wi), "wo": lin(l.mlp.Wo), "type": l.attention_type} for l in base.layers]
        if any(p["ln1"] is None for p in self.layers[1:]):
            raise ValueError("unexpected ModernBERT layout")
        self._rope = {}
        # the projections with a short reduction dimension through the streamed-weight kernel (ts_linear_act), as in
        # LeanBertEncoder; the MLP's down projection (K = intermediate size) stays a library GEMM
        self.fused_linear = True
        self.min_linear_rows = 4096
        for p in self.layers:
            p["tqkv"] = p["to"] = p["twi"] = None
            if compute_dtype in (torch.bfloat16, torch.float16) and p["qkv"][0].is_cuda:
                try:
                    from .index import TiledLinear
                    for key, src in (("tqkv", "qkv"), ("to", "o"), ("twi", "wi")):
                        w, b = p[src]
                        if TiledLinear.usable(int(w.shape[0]), int(w.shape[1])):
                            p[key] = TiledLinear(w, b)
                except Exception:
                    p["tqkv"] = p["to"] = p["twi"] = None

    def _tables(self, L: int, device):
        """cos / sin [L, head_dim] float32 per layer type, from the module's rotary embedding (positions 0..L-1)."""
        key = (L, str(device))
        if key not in self._rope:
            pos = torch.arange(L, device=device)[None, :]
            probe = torch.empty(0, dtype=torch.float32, device=device)
            self._rope[key] = {t: tuple(v[0].float().contiguous() for v in self.rotary(probe, pos, t))
                               for t in {p["type"] for p in self.layers}}
        return self._rope[key]

    @torch.no_grad()
    def hidden(self, input_ids, attention_mask, token_type_ids=None, lengths=None):
        cd = self.cd or torch.float32
        B, L = input_ids.shape
        H, nh = int(self.tok.shape[-1]), self.heads
        dh = H // nh
        dev = input_ids.device
        fused = (self.fused and self.cd in (torch.bfloat16, torch.float16) and input_ids.is_cuda and H % 4 == 0 and H <= 2048
                 and dh in (32, 64) and B <= 65535 and L <= (1120 if dh == 32 else 576))
        lens = None
        if fused:
            from .index import add_layernorm, attention_varlen, geglu
            lens = (lengths.to(torch.int32) if lengths is not None
                    else torch.full((B,), L, dtype=torch.int32, device=dev) if attention_mask is None
                    else attention_mask.sum(1, dtype=torch.int32))
            if lengths is None and attention_mask is not None and not bool(
                    (attention_mask.to(torch.bool) == (torch.arange(L, device=dev)[None, :] < lens[:, None])).all()):
                fused, lens = False, None           # holes or left padding: torch ops below

        def norm(new, old, ln, pre):
            """pre: (old + new, LayerNorm of it in the compute dtype); else (LayerNorm(old + new) fp32, its copy)."""
            if fused:
                return add_layernorm(new, old, ln[0], ln[1], ln[2], lp_dtype=cd, prenorm=pre)
            t = new + old if old is not None else new
            y = F.layer_norm(t.float(), (H,), ln[0], ln[1], ln[2])
            return (t, y.to(cd)) if pre else (y, y.to(cd))
        x, xb = norm(self.tok[input_ids], None, self.emb_ln, False)
        tables = self._tables(L, dev)
        masks = {}
        if not fused:
            valid = attention_mask.to(torch.bool) if attention_mask is not None else torch.ones((B, L), dtype=torch.bool, device=dev)
            t = torch.arange(L, device=dev)
            near = (t[:, None] - t[None, :]).abs() <= self.window
            masks = {"full_attention": valid[:, None, None, :], "sliding_attention": valid[:, None, None, :] & near[None, None]}
        else:
            abuf = torch.zeros((B, L, H), dtype=cd, device=dev)
        n = len(self.layers)
        tl = fused and self.fused_linear and B * L >= self.min_linear_rows
        for i, p in enumerate(self.layers):
            qkv = p["tqkv"](xb) if tl and p["tqkv"] is not None else F.linear(xb, *p["qkv"])
            cos, sin = tables[p["type"]]
            if fused:
                a = attention_varlen(qkv, lens, nh, out=abuf, window=self.window if p["type"] == "sliding_attention" else 0,
                                     rope=(cos, sin))          # rotary embedding applied as q and k are loaded
            else:
                q, k, v = (qkv.view(B, L, 3, nh, dh)[:, :, j].transpose(1, 2) for j in range(3))
                half = lambda z: torch.cat((-z[..., dh // 2:], z[..., : dh // 2]), dim=-1)
                q, k = ((z.float() * cos[None, None]) + (half(z.float()) * sin[None, None]) for z in (q, k))
                a = F.scaled_dot_product_attention(q.to(cd), k.to(cd), v, attn_mask=masks[p["type"]]).transpose(1, 2).reshape(B, L, H)
            o = p["to"](a) if tl and p["to"] is not None and a.is_contiguous() else F.linear(a, *p["o"])
            x, hb = norm(o, x, p["ln2"], True)
            u = p["twi"](hb) if tl and p["twi"] is not None else F.linear(hb, *p["wi"])
            if fused:
                g = geglu(u)
            else:
                inp, gate = u.chunk(2, dim=-1)
                g = F.gelu(inp) * gate
            f = F.linear(g, *p["wo"])
            if i + 1 < n:
                x, xb = norm(f, x, self.layers[i + 1]["ln1"], True)
            else:
                x, xb = norm(f, x, self.final_ln, False)
        return x, xb

    def __call__(self, input_ids, attention_mask, token_type_ids=None, lengths=None) -> torch.Tensor:
        return self.hidden(input_ids, attention_mask, token_type_ids, lengths)[0]


def lean_encoder_for(model, compute_dtype):
    """The LeanBertEncoder of a bare encoder module for one compute dtype (built once, kept on the module), or False
    when the architecture has none."""
    cache = model.__dict__.setdefault("_ts_lean_encoders", {})
    if compute_dtype not in cache:
        kind = getattr(getattr(model, "config", None), "model_type", None)
        try:
            cache[compute_dtype] = (LeanModernBertEncoder if kind == "modernbert" else LeanBertEncoder)(model, compute_dtype)
        except Exception:
            cache[compute_dtype] = False
    return cache[compute_dtype]


# --------------------------------------------------------------------------- cross-encoder
class CrossEncoderModel:
    """Stand-in for sentence_transformers.CrossEncoder (see module docstring)."""

    def __init__(self, model_name: str, device: str = "auto", max_length: int = 256,
                 cache_folder: str = "./models", amp_dtype=torch.bfloat16, use_amp: bool = True,
                 seed: int = 0, use_hip_graph: bool = False):
        self.device = resolve_device(device)
        self.use_hip_graph = use_hip_graph
        self._graphed: Optional[GraphedClassifier] = None
        self.tokenizer, self.model, _ = load_backbone(model_name, cache_folder, "seqcls", num_labels=1,
                                                      seed=seed)
        self.model.to(self.device).eval()
        self.max_length = int(max_length)
        self.num_labels = int(getattr(self.model.config, "num_labels", 1))
        act = getattr(self.model.config, "sbert_ce_default_activation_function", None)
        if act is not None:
            self.activation = "identity" if str(act).endswith("Identity") else "sigmoid"
        else:
            self.activation = "sigmoid" if self.num_labels == 1 else "identity"
        self.amp_dtype = amp_dtype
        self.use_amp = use_amp
        self.lean_forward = True      # batched reranking runs the written-out forward (LeanBertClassifier) when it applies
        self._lean: Any = None

    def _lean_model(self):
        if self._lean is None:
            try:
                amp = self.use_amp and str(self.device).startswith("cuda")
                self._lean = LeanBertClassifier(self.model, self.amp_dtype if amp else None)
            except Exception:
                self._lean = False    # another architecture: the transformers forward stays
        return self._lean

    @torch.no_grad()
    def logits(self, pairs: Sequence[Sequence[str]], batch_size: int = 32) -> torch.Tensor:
        """Raw logits [P, num_labels] (float32, on the model's device), batched in
        length-sorted order like CrossEncoder.predict."""
        pairs = [list(p) for p in pairs]
        order = np.argsort([-(len(p[0]) + len(p[1])) for p in pairs], kind="stable")
        if not pairs:
            return torch.zeros((0, self.num_labels), device=self.device)
        res = torch.empty((len(pairs), self.num_labels), dtype=torch.float32, device=self.device)
        if self.use_hip_graph and str(self.device).startswith("cuda") and len(pairs) <= GraphedClassifier.ROWS[-1]:
            # one query's pairs: a single forward replayed from a HIP graph
            enc = self.tokenizer([p[0] for p in pairs], [p[1] for p in pairs], truncation=True, padding=True,
                                 max_length=self.max_length, return_tensors="pt")
            enc = {k: to_device_async(v, self.device) for k, v in enc.items() if k in ("input_ids", "attention_mask", "token_type_ids")}
            lg = self.logits_graphed(enc)
            if lg is not None:
                return lg.reshape(len(pairs), -1)
        for s in range(0, len(pairs), batch_size):
            idx = order[s:s + batch_size]
            enc = self.tokenizer([pairs[i][0] for i in idx], [pairs[i][1] for i in idx], truncation=True,
                                 padding=True, max_length=self.max_length, return_tensors="pt")
            lengths = _prefix_lengths(enc, self.tokenizer)
            enc = {k: to_device_async(v, self.device) for k, v in enc.items() if k in ("input_ids", "attention_mask", "token_type_ids")}
            if lengths is not None:
                enc["lengths"] = to_device_async(lengths, self.device)
            lg = self.logits_from_ids(enc)        # the written-out forward when there is one, else the module under AMP
            res[to_device_async(torch.from_numpy(np.ascontiguousarray(idx)), self.device)] = lg.reshape(len(idx), -1)
        return res

    def logits_graphed(self, enc: Dict[str, torch.Tensor]) -> Optional[torch.Tensor]:
        """Logits [P, num_labels] of one query's assembled pairs from a replayed HIP graph, or None (graphs off, no
        bucket for the shape, capture failed: the caller runs the eager forward)."""
        if not (self.use_hip_graph and str(self.device).startswith("cuda")) or enc["input_ids"].shape[0] > GraphedClassifier.ROWS[-1]:
            return None
        if self._graphed is None:
            self._graphed = GraphedClassifier(self.model, getattr(self.tokenizer, "pad_token_id", 0) or 0,
                                              self.amp_dtype if self.use_amp else None,
                                              use_token_types="token_type_ids" in enc and
                                              hasattr(self.model.config, "type_vocab_size"))
        lg = self._graphed(enc)
        return lg.reshape(enc["input_ids"].shape[0], -1) if lg is not None else None

    def activate(self, lg: torch.Tensor) -> torch.Tensor:
        if self.activation == "sigmoid":
            lg = torch.sigmoid(lg)
        if self.num_labels == 1:
            lg = lg.squeeze(-1)
        return lg

    def predict(self, sentences: Sequence[Sequence[str]], batch_size: int = 32,
                show_progress_bar: bool = False, **_) -> np.ndarray:
        return self.activate(self.logits(sentences, batch_size=batch_size)).cpu().numpy().astype(np.float32)

    def packed_ok(self, max_len: int, n_seq: int) -> bool:
        """Whether logits_from_ids takes PairAssembler.batch_packed batches (the written-out forward on the GPU)."""
        lean = self._lean_model() if self.lean_forward else False
        return bool(lean) and lean.packed_ok(self.device, max_len, n_seq)

    @torch.no_grad()
    def logits_from_ids(self, enc: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Raw logits [P, num_labels] (float32) for already assembled id tensors on the model's device."""
        if "token_type_ids" in enc and not hasattr(self.model.config, "type_vocab_size"):
            enc = {k: v for k, v in enc.items() if k != "token_type_ids"}
        lean = self._lean_model() if self.lean_forward else False
        if enc.get("packed"):
            if not lean:
                raise ValueError("packed batches need the written-out forward (see packed_ok)")
            types = enc.get("token_type_ids") if hasattr(self.model.config, "type_vocab_size") else None
            return lean.logits_packed(enc["input_ids"], enc["positions"], types, enc["lengths"], enc["offsets"],
                                      enc["max_len"]).reshape(enc["lengths"].shape[0], -1)
        if lean:
            return lean(enc["input_ids"], enc["attention_mask"], enc.get("token_type_ids"),
                        lengths=enc.get("lengths")).reshape(enc["input_ids"].shape[0], -1)
        enc = {k: v for k, v in enc.items() if k != "lengths"}
        with _autocast(self.device, self.use_amp, self.amp_dtype):
            return self.model(**enc).logits.float().reshape(enc["input_ids"].shape[0], -1)
