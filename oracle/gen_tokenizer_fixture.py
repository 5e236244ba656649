#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not shipped): builds the tokenizer fixture and its golden vectors.

The reference tokenizes with HF `tokenizers` (Rust) through mlc-ai/tokenizers-cpp
(/root/reference/src/tokenizer.c:33 `tokenizers_encode_batch`, :175 `tokenizers_new_from_str`) on the model
repo's `tokenizer/tokenizer.json` (/root/reference/include/paths.h:4) -- a file that is not on disk here.  The
python `tokenizers` wheel in this image (0.22.2) is the same Rust library, so it is the oracle for the native
tokenizer (gliclass/c_amd/host/tokenizer.c):

1. a SentencePiece unigram model is trained offline on text available in this image (python docstrings + a
   hand-written multilingual sample), which also yields the real `nmt_nfkc` precompiled character map;
2. it is converted the way transformers' DebertaV2Converter does it
   (transformers/convert_slow_tokenizer.py: Strip -> Precompiled -> Replace(" {2,}"," "), Metaspace, Unigram,
   TemplateProcessing "[CLS] $A [SEP]"), the GLiClass tokens <<LABEL>> / <<SEP>> are added
   (/root/reference/src/preprocessor.c:68-69) and `tokenizer.json` is saved -- the same structure as
   microsoft/deberta-v3-*'s file;
3. `tokenizers` encodes a set of probe texts (prompts built like the reference does, multilingual, odd
   whitespace, combining marks, unknown characters, empty strings ...) -> tests/golden/tokenizer_golden.json.

Run here (needs sentencepiece + tokenizers + transformers; no network):  python oracle/gen_tokenizer_fixture.py
"""
import gzip
import inspect
import json
import os
import pydoc
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

MULTI = [
    "Der schnelle braune Fuchs springt über den faulen Hund. Äpfel, Öl und Übermut tun selten gut – größer als gedacht.",
    "Portez ce vieux whisky au juge blond qui fume. L'élève a été reçu à l'école; ça coûte très cher, n'est-ce pas ?",
    "El veloz murciélago hindú comía feliz cardillo y kiwi. ¿Dónde está la cigüeña? ¡Mañana será otro día!",
    "Съешь же ещё этих мягких французских булок, да выпей чаю. Широкая электрификация южных губерний даст мощный толчок.",
    "Ξεσκεπάζω την ψυχοφθόρα βδελυγμία. Η γρήγορη καφέ αλεπού πηδάει πάνω από το τεμπέλικο σκυλί.",
    "敏捷的棕色狐狸跳过了懒狗。今天天气很好，我们去公园散步吧。人工智能正在改变世界。",
    "いろはにほへと ちりぬるを。素早い茶色の狐はのろまな犬を飛び越える。東京は日本の首都です。カタカナとひらがな。",
    "다람쥐 헌 쳇바퀴에 타고파. 빠른 갈색 여우가 게으른 개를 뛰어넘습니다. 서울은 대한민국의 수도입니다.",
    "نص حكيم له سر قاطع وذو شأن عظيم مكتوب على ثوب أخضر ومغلف بجلد أزرق. الثعلب البني السريع يقفز فوق الكلب الكسول.",
    "ऋषियों को सताने वाले दुष्ट राक्षसों के राजा रावण का सर्वनाश करने वाले विष्णुवतार भगवान श्रीराम।",
    "Pchnąć w tę łódź jeża lub ośm skrzyń fig. Zażółć gęślą jaźń. Příliš žluťoučký kůň úpěl ďábelské ódy.",
    "Pijamalı hasta yağız şoföre çabucak güvendi. Árvíztűrő tükörfúrógép. Flygande bäckasiner söka hwila på mjuka tuvor.",
    "The ﬁrst ﬂight cost ½ of the price: ① ② ③, ℃ and ㎞, ＦＵＬＬＷＩＤＴＨ text, ｈａｌｆ ｶﾀｶﾅ, x² + y³ ≠ z™ … “quoted” ‘single’ — dash.",
    "Emoji: 😀 👍🏽 👨‍👩‍👧‍👦 🇺🇦 ❤️ and maths ∑ ∫ √ ∞ ≈ plus currency € £ ¥ ₹ ₿.",
]


def english_corpus():
    mods = ["os", "sys", "re", "json", "collections", "itertools", "functools", "subprocess", "threading", "socket", "email",
            "http.client", "urllib.request", "argparse", "logging", "unittest", "asyncio", "typing", "decimal", "datetime",
            "pathlib", "shutil", "sqlite3", "xml.dom.minidom", "csv", "random", "statistics", "string", "textwrap", "heapq",
            "numpy", "numpy.linalg", "numpy.fft", "scipy.optimize", "scipy.signal", "pandas"]
    out = []
    for m in mods:
        try:
            mod = __import__(m, fromlist=["x"])
            txt = pydoc.render_doc(mod, renderer=pydoc.plaintext)
        except Exception:
            continue
        for line in txt.splitlines():
            line = line.strip(" |")
            if len(line) > 20:
                out.append(line)
    return out


def build_tokenizer(workdir, vocab_size=6000):
    import sentencepiece as spm
    from sentencepiece import sentencepiece_model_pb2 as pb
    from tokenizers import AddedToken, Regex, Tokenizer, decoders, normalizers, pre_tokenizers, processors
    from tokenizers.models import Unigram

    corpus = os.path.join(workdir, "corpus.txt")
    eng = english_corpus()
    rnd = random.Random(0)
    rnd.shuffle(eng)
    eng = eng[:60000]
    with open(corpus, "w", encoding="utf-8") as f:
        for line in eng:
            f.write(line + "\n")
        for _ in range(40):
            for line in MULTI:
                f.write(line + "\n")
    # DeBERTa-v3's spm.model layout: [PAD]=0 [CLS]=1 [SEP]=2 [UNK]=3, pieces, [MASK] last
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(workdir, "spm"), vocab_size=vocab_size,
                                   model_type="unigram", character_coverage=0.9995, num_threads=1,
                                   pad_id=0, pad_piece="[PAD]", bos_id=1, bos_piece="[CLS]", eos_id=2, eos_piece="[SEP]",
                                   unk_id=3, unk_piece="[UNK]", normalization_rule_name="nmt_nfkc",
                                   input_sentence_size=200000, shuffle_input_sentence=False, minloglevel=2)
    m = pb.ModelProto()
    m.ParseFromString(open(os.path.join(workdir, "spm.model"), "rb").read())
    vocab = [(p.piece, p.score) for p in m.pieces]
    vocab.append(("[MASK]", 0.0))
    tok = Tokenizer(Unigram(vocab, unk_id=3, byte_fallback=False))
    tok.normalizer = normalizers.Sequence([normalizers.Strip(), normalizers.Precompiled(m.normalizer_spec.precompiled_charsmap),
                                           normalizers.Replace(Regex(" {2,}"), " ")])
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="always")])
    tok.decoder = decoders.Metaspace(replacement="▁", prepend_scheme="always")
    tok.post_processor = processors.TemplateProcessing(single="[CLS]:0 $A:0 [SEP]:0", pair="[CLS]:0 $A:0 [SEP]:0 $B:1 [SEP]:1",
                                                       special_tokens=[("[CLS]", 1), ("[SEP]", 2)])
    tok.add_special_tokens(["[PAD]", "[CLS]", "[SEP]", "[UNK]", "[MASK]"])
    tok.add_tokens(["<<LABEL>>", "<<SEP>>"])              # what `tokenizer.add_tokens([...])` upstream produces
    return tok


def probe_texts():
    sys.path.insert(0, ROOT)
    texts = [
        "", " ", "   \t\n ", "a", "Hello world", "  leading and trailing   spaces  ", "multiple    spaces\tand\ttabs\nnewlines",
        "One day I will see the world!", "ONNX is an open-source format designed to enable the interoperability of AI models.",
        "<<LABEL>>travel<<LABEL>>dreams<<LABEL>>sport<<LABEL>>science<<LABEL>>politics<<SEP>>One day I will see the world!",
        "One day I will see the world!<<LABEL>>travel<<LABEL>>dreams<<SEP>>",
        "<<LABEL>> spaced label <<LABEL>>x<<SEP>> text", "<<LABEL>><<LABEL>><<SEP>>", "<<LABEL", "a<<SEP>>b<<SEP>>", "<<label>>lower<<sep>>",
        "[CLS] literal specials [SEP] [MASK] [UNK] [PAD] inside", "x[SEP]y", "[ CLS ]",
        "é vs é; ǟ stacked; ́ lone mark", "ｆｕｌｌ ｗｉｄｔｈ １２３ ﬁ ﬂ ½ ™ ℃ ㎞ ① Ⅻ", "nbsp here em​zero‍width﻿bom",
        "control\x01chars\x7f and  nel   ls", "tab\tseparated\tvalues", "ümlaut Ünïcödé naïve façade", "日本語のテキストと中文文本和한국어 텍스트",
        "ﾊﾝｶｸ ｶﾀｶﾅ ﾃﾞｽ", "가각갂 각 jamo", "😀 👍🏽 👨‍👩‍👧‍👦 🇺🇦 ❤️ keycap 1️⃣", "𝔘𝔫𝔦𝔠𝔬𝔡𝔢 𝒎𝒂𝒕𝒉 𝟘𝟙𝟚", "Ǆ ǅ ǆ ß ẞ ſ ı İ",
        "العربية لُغَة ﷺ ﻻ", "क्षत्रिय ज़िंदगी श्रीमान्", "ไทย ภาษา กำลัง", "a" * 300, "word " * 200,
        "supercalifragilisticexpialidocious antidisestablishmentarianism", "3.14159 2,718 1e-7 0x7fff 100% $5.00 #tag @user http://example.com/a?b=c&d=e",
        "▁already metaspace▁inside", "quote “curly” ‘single’ «guillemets» – — … •", "­ soft­hyphen", "ⓐⓑⓒ ㈱ ㍿ ㌔ ㍍",
    ] + MULTI
    rnd = random.Random(7)
    labels = ["travel", "dreams", "sport", "science", "politics", "Machine Learning", "économie", "健康", "Искусство", "world news"]
    for i in range(24):                              # prompts in the reference's two layouts (src/preprocessor.c:84-108), labels lower-cased
        k = rnd.randint(1, 8)
        ls = [l.lower() if l.isascii() else l for l in rnd.sample(labels, k)]
        body = " ".join(rnd.choice(MULTI + texts[7:9]) for _ in range(rnd.randint(1, 4)))
        p = "".join("<<LABEL>>" + l for l in ls) + "<<SEP>>"
        texts.append(p + body if i % 2 == 0 else body + p)
    pools = [(0x20, 0x7f), (0xa0, 0x250), (0x300, 0x370), (0x370, 0x400), (0x400, 0x500), (0x590, 0x700), (0x900, 0x980),
             (0xe00, 0xe80), (0x1100, 0x1200), (0x1e00, 0x2000), (0x2000, 0x2070), (0x2100, 0x2200), (0x2460, 0x2500),
             (0x3000, 0x3100), (0x3300, 0x3400), (0x4e00, 0x4f00), (0xac00, 0xad00), (0xfb00, 0xfb50), (0xfe00, 0xfe10),
             (0xff00, 0xfff0), (0x1f1e6, 0x1f200), (0x1f300, 0x1f650), (0x1d400, 0x1d500), (0xe0020, 0xe0080)]
    for i in range(160):                             # random code-point soup: normaliser + grapheme segmentation + unknowns
        n = rnd.randint(1, 40)
        s = []
        for _ in range(n):
            lo, hi = rnd.choice(pools) if rnd.random() < 0.7 else pools[0]
            c = rnd.randrange(lo, hi)
            if 0xd800 <= c < 0xe000:
                c = 0x41
            s.append(chr(c))
            if rnd.random() < 0.15:
                s.append(rnd.choice(["́", "̈", "‍", "️", " ", "  ", "゙", "\U0001f3fd", "्"]))
        texts.append("".join(s))
    return texts


def main():
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as wd:
        tok = build_tokenizer(wd)
        js = tok.to_str()
    from tokenizers import Tokenizer
    j2 = json.loads(js)
    for a in j2["added_tokens"]:
        if a["content"] in ("<<LABEL>>", "<<SEP>>"):
            a.update(normalized=False, special=True)
    tok_sp = Tokenizer.from_str(json.dumps(j2))
    with gzip.GzipFile(os.path.join(GOLD, "tokenizer.json.gz"), "wb", mtime=0) as f:
        f.write(js.encode("utf-8"))
    texts = probe_texts()
    norm = [tok.normalizer.normalize_str(t) for t in texts]
    enc = [tok.encode(t, add_special_tokens=True).ids for t in texts]
    enc_nospecial = [tok.encode(t, add_special_tokens=False).ids for t in texts]
    # variant: <<LABEL>>/<<SEP>> registered as *special* (not normalised, matched on the raw text) -- only the added_tokens
    # block differs, so the test patches it into the same tokenizer.json
    added_special = json.loads(tok_sp.to_str())["added_tokens"]
    enc_sp = [tok_sp.encode(t, add_special_tokens=True).ids for t in texts]
    gold = {"generator": "oracle/gen_tokenizer_fixture.py", "tokenizers_version": __import__("tokenizers").__version__,
            "texts": texts, "normalized": norm, "ids": enc, "ids_no_special": enc_nospecial,
            "added_tokens_special_variant": added_special, "ids_special_variant": enc_sp}
    with gzip.GzipFile(os.path.join(GOLD, "tokenizer_golden.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(gold, ensure_ascii=True).encode("ascii"))
    print("vocab", tok.get_vocab_size(), "texts", len(texts), "json bytes", len(js))


if __name__ == "__main__":
    main()
