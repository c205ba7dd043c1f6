"""ORACLE -- test infrastructure only.

Independent restatement of the character path of the reference's text handling:
  encode_char              <- /root/reference/retokenize.py:5-17  (aligned_unit_type == 'char')
  split_tokens_on_spaces   <- /root/reference/retokenize.py:19-39 (char branch) on top of the upstream
                              whisper.tokenizer.Tokenizer.split_tokens_on_unicode (restated)
  CharTokenizer            <- the subset of whisper.tokenizer.Tokenizer the char path touches, with the
                              GPT-2 byte -> rank table (' ' -> 220, 'a' -> 64) and the multilingual specials.
Pinned against the real retokenize.py (run under a num2words stub) by tests/golden/make_golden.py.
"""
import string


def _byte_to_rank():
    printable = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    others = [b for b in range(256) if b not in printable]
    return {b: i for i, b in enumerate(printable + others)}


class CharTokenizer:
    """Multilingual Whisper tokenizer restricted to single-byte tokens and special tokens."""

    def __init__(self):
        self.b2r = _byte_to_rank()
        self.r2b = {r: b for b, r in self.b2r.items()}
        self.eot = 50257
        self.sot = 50258
        self.sot_sequence = (50258, 50259, 50359)
        self.no_timestamps = 50363
        self.timestamp_begin = 50364

    def encode(self, text):
        return [self.b2r[b] for b in text.encode("utf-8")]

    def decode_with_timestamps(self, tokens):
        out = b""
        for t in tokens:
            if t == self.eot:
                out += b"<|endoftext|>"
            elif t >= self.timestamp_begin:
                out += ("<|%.2f|>" % ((t - self.timestamp_begin) * 0.02)).encode()
            elif t in self.r2b:
                out += bytes([self.r2b[t]])
            else:
                out += ("<|special_%d|>" % t).encode()
        return out.decode("utf-8", errors="replace")

    def split_tokens_on_unicode(self, tokens):
        full = self.decode_with_timestamps(tokens)
        bad = "�"
        words, word_tokens, cur, offset = [], [], [], 0
        for t in tokens:
            cur.append(t)
            dec = self.decode_with_timestamps(cur)
            if bad not in dec or full[offset + dec.index(bad)] == bad:
                words.append(dec)
                word_tokens.append(cur)
                cur = []
                offset += len(dec)
        return words, word_tokens

    def split_to_word_tokens(self, tokens):
        """whisper.tokenizer.Tokenizer.split_to_word_tokens for a space-delimited language (upstream
        split_tokens_on_spaces, restated from its published algorithm): a new word starts at a special
        token, at a piece that starts with a space, at a piece that is pure punctuation, or first; other pieces are
        appended to the current word. Used by the reference's default_find_alignment (timing.py:167)."""
        subwords, subword_tokens_list = self.split_tokens_on_unicode(tokens)
        words, word_tokens = [], []
        for subword, subword_tokens in zip(subwords, subword_tokens_list):
            special = subword_tokens[0] >= self.eot
            with_space = subword.startswith(" ")
            punctuation = subword.strip() in string.punctuation
            if special or with_space or punctuation or len(words) == 0:
                words.append(subword)
                word_tokens.append(subword_tokens)
            else:
                words[-1] = words[-1] + subword
                word_tokens[-1].extend(subword_tokens)
        return words, word_tokens


def encode_char(text, tokenizer):
    tokens = []
    space_id = tokenizer.encode(" ")
    wrds = text.split()
    for i, w in enumerate(wrds):
        for c in w:
            tokens += tokenizer.encode(c)
        if i < len(wrds) - 1:
            tokens += space_id
    return tokens


def split_tokens_on_spaces(tokens, tokenizer, aligned_unit_type="char"):
    assert aligned_unit_type == "char", "the oracle restates the char path only"
    subwords, subword_tokens_list = tokenizer.split_tokens_on_unicode(tokens)
    words, word_tokens = [], []
    for subword, subword_tokens in zip(subwords, subword_tokens_list):
        special = subword_tokens[0] >= tokenizer.eot
        with_space = subword == " "
        if special or with_space or len(words) == 0:
            words.append(subword)
            word_tokens.append(subword_tokens)
        else:
            words[-1] = words[-1] + subword
            word_tokens[-1].extend(subword_tokens)
    return words, word_tokens
