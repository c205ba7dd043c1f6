"""Text <-> token helpers with the reference's names and behaviour (retokenize.py:5-50):
`encode` (char / subword), `split_tokens_on_spaces` (the word-boundary merge used by force_align,
timing.py:105) and `remove_punctuation`. Pure host-side string work."""
import string

_KEEP_APOSTROPHE = str.maketrans("", "", string.punctuation.replace("'", ""))

_ONES = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve",
         "thirteen", "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_TENS = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
_SCALES = [(10 ** 18, "quintillion"), (10 ** 15, "quadrillion"), (10 ** 12, "trillion"), (10 ** 9, "billion"),
           (10 ** 6, "million"), (10 ** 3, "thousand")]


def _below_thousand(n):
    parts = []
    if n >= 100:
        parts.append(_ONES[n // 100] + " hundred")
        n %= 100
        if n:
            parts.append("and")
    if n >= 20:
        parts.append(_TENS[n // 10] + ("-" + _ONES[n % 10] if n % 10 else ""))
    elif n > 0 or not parts:
        parts.append(_ONES[n])
    return " ".join(parts)


def number_to_words(n):
    """English cardinal in the num2words style ('one thousand, two hundred and thirty-four'); used when the
    optional `num2words` package (retokenize.py:2 of the reference) is not installed."""
    try:
        from num2words import num2words
        return num2words(n)
    except ImportError:
        pass
    if n < 0:
        return "minus " + number_to_words(-n)
    if n < 1000:
        return _below_thousand(n)
    groups = []
    rest = n
    for value, name in _SCALES:
        if rest >= value:
            groups.append(_below_thousand(rest // value) + " " + name)
            rest %= value
    if rest:
        tail = _below_thousand(rest)
        if rest < 100:
            return ", ".join(groups) + " and " + tail
        groups.append(tail)
    return ", ".join(groups)


def encode(text, tokenizer, aligned_unit_type="subword"):
    """subword: tokenizer.encode(text). char: every character of every word encoded on its own, with one
    encoded ' ' between consecutive words (retokenize.py:9-17)."""
    assert aligned_unit_type in ["char", "subword"]
    if aligned_unit_type == "subword":
        return tokenizer.encode(text)
    space = tokenizer.encode(" ")
    out = []
    words = text.split()
    for wi, word in enumerate(words):
        if wi > 0:
            out += space
        for ch in word:
            out += tokenizer.encode(ch)
    return out


def split_tokens_on_spaces(tokens, tokenizer, aligned_unit_type="subword"):
    """Groups tokens into words. char mode (retokenize.py:24-37): a new word starts at the first piece,
    at every special token (id >= eot) and at every piece that is exactly ' ' (which then owns the
    following letters); everything else is appended to the current word."""
    assert aligned_unit_type in ["char", "subword"]
    if aligned_unit_type == "subword":
        return tokenizer.split_to_word_tokens(tokens)
    pieces, piece_tokens = tokenizer.split_tokens_on_unicode(tokens)
    words, word_tokens = [], []
    for piece, toks in zip(pieces, piece_tokens):
        starts_word = (not words) or toks[0] >= tokenizer.eot or piece == " "
        if starts_word:
            words.append(piece)
            word_tokens.append(toks)
        else:
            words[-1] += piece
            word_tokens[-1].extend(toks)
    return words, word_tokens


def char_word_starts(tokens, tokenizer):
    """Vectorised equivalent of split_tokens_on_spaces(..., 'char') for the common case where every
    non-special token is a single ASCII byte: returns the index of the first token of every word
    (== np.pad(np.cumsum(len(word_tokens[:-1])), (1, 0)) plus the start of the last word), or None
    when the fast path does not apply (multi-byte characters) and the caller must use the general splitter."""
    import numpy as np
    t = np.asarray(tokens, dtype=np.int64)
    if t.size == 0:
        return np.zeros(0, dtype=np.int64)
    special = t >= tokenizer.eot
    tab = tokenizer.single_ascii_table
    plain = ~special
    if not bool(np.all(tab[np.where(plain, t, 0)] | special)):
        return None
    space = tokenizer.encode(" ")
    starts = special.copy()
    if len(space) == 1:
        starts |= (t == space[0])
    starts[0] = True
    return np.flatnonzero(starts)


def remove_punctuation(text):
    """Drops punctuation except apostrophes and spells out all-digit words (retokenize.py:41-50)."""
    text = text.translate(_KEEP_APOSTROPHE)
    words = []
    for w in text.split():
        if w.isdigit():
            w = number_to_words(int(w))
        words.append(w.strip(string.punctuation))
    return " ".join(words).translate(_KEEP_APOSTROPHE)
