"""ctypes view of the native tokenizer (include/tokenizers_c.h, include/tokenizer.h) for tests and scripts."""
import ctypes as C

from . import _lib

_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


class Tokenizer:
    def __init__(self, json_text):
        if isinstance(json_text, str):
            json_text = json_text.encode("utf-8")
        self._L = _lib.model()
        self._h = self._L.tokenizers_new_from_str(json_text, len(json_text))
        if not self._h:
            raise ValueError("tokenizers_new_from_str failed (see stderr)")

    @classmethod
    def from_file(cls, path):
        self = cls.__new__(cls)
        self._L = _lib.model()
        self._h = self._L.create_tokenizer(path.encode())
        if not self._h:
            raise ValueError(f"create_tokenizer({path}) failed (see stderr)")
        return self

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self._L.tokenizers_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, text, add_special_tokens=True):
        b = text.encode("utf-8") if isinstance(text, str) else text
        r = _lib.TokenizerEncodeResult()
        self._L.tokenizers_encode(self._h, b, len(b), int(add_special_tokens), C.byref(r))
        ids = [r.token_ids[i] for i in range(r.len)]
        _libc.free(r.token_ids)
        return ids

    def encode_batch(self, texts, add_special_tokens=True):
        bs = [t.encode("utf-8") if isinstance(t, str) else t for t in texts]
        n = len(bs)
        arr = (C.c_char_p * n)(*bs)
        lens = (C.c_size_t * n)(*[len(b) for b in bs])
        res = (_lib.TokenizerEncodeResult * n)()
        self._L.tokenizers_encode_batch(self._h, arr, lens, n, int(add_special_tokens), res)
        out = [[res[i].token_ids[j] for j in range(res[i].len)] for i in range(n)]
        self._L.tokenizers_free_encode_results(res, n)
        return out

    def normalize(self, text):
        b = text.encode("utf-8") if isinstance(text, str) else text
        n = C.c_size_t()
        p = self._L.glc_tokenizer_normalize(self._h, b, len(b), C.byref(n))
        s = C.string_at(p, n.value).decode("utf-8")
        _libc.free(p)
        return s

    def decode(self, ids, skip_special_tokens=True):
        arr = (C.c_uint32 * len(ids))(*ids)
        self._L.tokenizers_decode(self._h, arr, len(ids), int(skip_special_tokens))
        p, n = C.c_char_p(), C.c_size_t()
        self._L.tokenizers_get_decode_str(self._h, C.byref(p), C.byref(n))
        return C.string_at(p, n.value).decode("utf-8")

    def vocab_size(self):
        n = C.c_size_t()
        self._L.tokenizers_get_vocab_size(self._h, C.byref(n))
        return n.value

    def id_to_token(self, i):
        p, n = C.c_void_p(), C.c_size_t()
        self._L.tokenizers_id_to_token(self._h, i, C.byref(p), C.byref(n))
        return C.string_at(p, n.value).decode("utf-8") if n.value else ""

    def token_to_id(self, tok):
        b = tok.encode("utf-8")
        i = C.c_int32()
        self._L.tokenizers_token_to_id(self._h, b, len(b), C.byref(i))
        return i.value

    def tokenize_inputs(self, texts, max_length):
        """The reference's tokenize_inputs(): returns (ids [B,S], mask [B,S]) as nested lists."""
        bs = [t.encode("utf-8") for t in texts]
        arr = (C.c_char_p * len(bs))(*bs)
        t = self._L.tokenize_inputs(self._h, arr, len(bs), max_length)
        ids = [[t.input_ids[i][j] for j in range(t.seq_length)] for i in range(t.batch_size)]
        mask = [[t.attention_mask[i][j] for j in range(t.seq_length)] for i in range(t.batch_size)]
        tt = [[t.token_type_ids[i][j] for j in range(t.seq_length)] for i in range(t.batch_size)]
        assert all(v == 0 for row in tt for v in row)
        self._L.free_tokenized_inputs(C.byref(t))
        return ids, mask
