"""Tokenizer handle: what callers pass as `tokenizer` (reference: whisper.tokenizer.get_tokenizer at
infer_ali.py:41 / README.md:95). Exposes the members the reference touches: `sot_sequence`,
`no_timestamps`, `eot`, `encode`, `decode` (plot.py:52), `split_tokens_on_unicode` (retokenize.py:24),
`split_to_word_tokens` (retokenize.py:22), `decode_with_timestamps`.

Character alignment over ASCII needs only the GPT-2 byte -> rank table, which is derivable offline
(' ' -> 220, 'a' -> 64). Sub-word mode and non-ASCII text need the real merge
table: pass `vocab_path=` pointing at a local tiktoken file (multilingual.tiktoken / gpt2.tiktoken);
nothing is ever fetched.
"""
import base64
import string
from functools import cached_property

LANGUAGES = {
    "en": "english", "zh": "chinese", "de": "german", "es": "spanish", "ru": "russian", "ko": "korean", "fr": "french",
    "ja": "japanese", "pt": "portuguese", "tr": "turkish", "pl": "polish", "ca": "catalan", "nl": "dutch", "ar": "arabic",
    "sv": "swedish", "it": "italian", "id": "indonesian", "hi": "hindi", "fi": "finnish", "vi": "vietnamese", "he": "hebrew",
    "uk": "ukrainian", "el": "greek", "ms": "malay", "cs": "czech", "ro": "romanian", "da": "danish", "hu": "hungarian",
    "ta": "tamil", "no": "norwegian", "th": "thai", "ur": "urdu", "hr": "croatian", "bg": "bulgarian", "lt": "lithuanian",
    "la": "latin", "mi": "maori", "ml": "malayalam", "cy": "welsh", "sk": "slovak", "te": "telugu", "fa": "persian",
    "lv": "latvian", "bn": "bengali", "sr": "serbian", "az": "azerbaijani", "sl": "slovenian", "kn": "kannada",
    "et": "estonian", "mk": "macedonian", "br": "breton", "eu": "basque", "is": "icelandic", "hy": "armenian", "ne": "nepali",
    "mn": "mongolian", "bs": "bosnian", "kk": "kazakh", "sq": "albanian", "sw": "swahili", "gl": "galician", "mr": "marathi",
    "pa": "punjabi", "si": "sinhala", "km": "khmer", "sn": "shona", "yo": "yoruba", "so": "somali", "af": "afrikaans",
    "oc": "occitan", "ka": "georgian", "be": "belarusian", "tg": "tajik", "sd": "sindhi", "gu": "gujarati", "am": "amharic",
    "yi": "yiddish", "lo": "lao", "uz": "uzbek", "fo": "faroese", "ht": "haitian creole", "ps": "pashto", "tk": "turkmen",
    "nn": "nynorsk", "mt": "maltese", "sa": "sanskrit", "lb": "luxembourgish", "my": "myanmar", "bo": "tibetan",
    "tl": "tagalog", "mg": "malagasy", "as": "assamese", "tt": "tatar", "haw": "hawaiian", "ln": "lingala", "ha": "hausa",
    "ba": "bashkir", "jw": "javanese", "su": "sundanese", "yue": "cantonese",
}
TO_LANGUAGE_CODE = {**{name: code for code, name in LANGUAGES.items()}, "burmese": "my", "valencian": "ca", "flemish": "nl",
                    "haitian": "ht", "letzeburgesch": "lb", "pushto": "ps", "panjabi": "pa", "moldavian": "ro", "moldovan": "ro",
                    "sinhalese": "si", "castilian": "es", "mandarin": "zh"}


def _byte_ranks():
    """GPT-2 byte ordering: printable bytes first (33..126, 161..172, 174..255), then the rest in order."""
    order = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    rest = [b for b in range(256) if b not in set(order)]
    return {bytes([b]): r for r, b in enumerate(order + rest)}


class NeedVocabError(RuntimeError):
    pass


class Tokenizer:
    def __init__(self, multilingual=True, num_languages=99, language="en", task="transcribe", vocab_path=None):
        self.multilingual = multilingual
        self.num_languages = num_languages
        self.language = language if multilingual else None
        self.task = task if multilingual else None
        self.has_vocab = vocab_path is not None
        if vocab_path is not None:
            ranks = {}
            with open(vocab_path, "rb") as f:
                for line in f:
                    if line.strip():
                        tok, rank = line.split()
                        ranks[base64.b64decode(tok)] = int(rank)
            self._ranks = ranks
        else:
            self._ranks = _byte_ranks()
        self._bytes_of = {r: b for b, r in self._ranks.items()}
        n_base = 50257 if multilingual else 50256  # size of the BPE vocabulary below the specials
        if self.has_vocab and len(self._ranks) != n_base:
            raise ValueError("vocab file has %d ranks, expected %d" % (len(self._ranks), n_base))
        specials = ["<|endoftext|>", "<|startoftranscript|>"]
        specials += ["<|%s|>" % code for code in list(LANGUAGES.keys())[:num_languages]]
        specials += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
        specials += ["<|%.2f|>" % (i * 0.02) for i in range(1501)]
        self.special_tokens = {s: n_base + i for i, s in enumerate(specials)}
        self._special_text = {v: k for k, v in self.special_tokens.items()}
        self.n_vocab = n_base + len(specials)

    # ---- special ids
    @cached_property
    def eot(self):
        return self.special_tokens["<|endoftext|>"]

    @cached_property
    def sot(self):
        return self.special_tokens["<|startoftranscript|>"]

    @cached_property
    def transcribe(self):
        return self.special_tokens["<|transcribe|>"]

    @cached_property
    def translate(self):
        return self.special_tokens["<|translate|>"]

    @cached_property
    def no_timestamps(self):
        return self.special_tokens["<|notimestamps|>"]

    @cached_property
    def timestamp_begin(self):
        return self.special_tokens["<|0.00|>"]

    @cached_property
    def sot_lm(self):
        return self.special_tokens["<|startoflm|>"]

    @cached_property
    def sot_prev(self):
        return self.special_tokens["<|startofprev|>"]

    @cached_property
    def no_speech(self):
        return self.special_tokens["<|nospeech|>"]

    @cached_property
    def non_speech_tokens(self):
        """whisper.tokenizer.Tokenizer.non_speech_tokens (upstream, restated): tokens of speaker tags / non-speech
        annotations that SuppressTokens removes. A symbol contributes when it (or " " + it) encodes to ONE token, the
        musical symbols always contribute their first token. Without a vocabulary file only single-byte symbols can
        be encoded, so the list then holds just those (enough for the byte-level dry-run vocabulary)."""
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』')
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♫♫".split()
        miscellaneous = set("♩♪♫♬♭♮♯")
        result = set()
        for first in (" -", " '"):
            try:
                result.add(self.encode(first)[0])
            except NeedVocabError:
                pass
        for symbol in symbols + list(miscellaneous):
            for text in (symbol, " " + symbol):
                try:
                    tokens = self.encode(text)
                except NeedVocabError:
                    continue
                if len(tokens) == 1 or symbol in miscellaneous:
                    result.add(tokens[0])
        return tuple(sorted(result))

    @cached_property
    def sot_sequence(self):
        seq = [self.sot]
        if self.language is not None:
            seq.append(self.sot + 1 + list(LANGUAGES.keys()).index(self.language))
        if self.task is not None:
            seq.append(self.transcribe if self.task == "transcribe" else self.translate)
        return tuple(seq)

    @cached_property
    def sot_sequence_including_notimestamps(self):
        return tuple(list(self.sot_sequence) + [self.no_timestamps])

    # ---- encode / decode
    def _bpe(self, piece):
        parts = [bytes([b]) for b in piece]
        while len(parts) > 1:
            best, best_rank = None, None
            for i in range(len(parts) - 1):
                r = self._ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best is None:
                break
            parts[best:best + 2] = [parts[best] + parts[best + 1]]
        return [self._ranks[p] for p in parts]

    def encode(self, text):
        data = text.encode("utf-8")
        if not self.has_vocab:
            if len(data) == 1:
                return [self._ranks[data]]
            if len(data) == 0:
                return []
            raise NeedVocabError(
                "encoding %r needs the BPE merge table: pass vocab_path=<local *.tiktoken file> to get_tokenizer "
                "(without it only single-byte characters can be encoded, which is all char alignment over ASCII needs)" % text)
        import regex
        pat = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""
        out = []
        for piece in regex.findall(pat, text):
            out.extend(self._bpe(piece.encode("utf-8")))
        return out

    def _token_bytes(self, t):
        if t in self._bytes_of:
            return self._bytes_of[t]
        if t in self._special_text:
            return self._special_text[t].encode()
        if not self.has_vocab:
            raise NeedVocabError("token id %d cannot be decoded without the BPE vocabulary: pass vocab_path=<local *.tiktoken file>" % t)
        raise KeyError("token id %d is neither in the loaded vocabulary nor a special token" % t)

    def decode(self, tokens):
        tokens = [t for t in tokens if t < self.timestamp_begin]
        return b"".join(self._token_bytes(int(t)) for t in tokens).decode("utf-8", errors="replace")

    def decode_with_timestamps(self, tokens):
        return b"".join(self._token_bytes(int(t)) for t in tokens).decode("utf-8", errors="replace")

    @cached_property
    def single_ascii_table(self):
        """bool[n_vocab]: token is exactly one ASCII byte (every such token is its own unicode piece)."""
        import numpy as np
        t = np.zeros(self.n_vocab, dtype=bool)
        for b, r in self._ranks.items():
            if len(b) == 1 and b[0] < 0x80:
                t[r] = True
        return t

    # ---- word splitting (whisper.tokenizer.Tokenizer.split_*)
    def split_tokens_on_unicode(self, tokens):
        decoded_full = self.decode_with_timestamps(tokens)
        replacement_char = "�"
        words, word_tokens, current, unicode_offset = [], [], [], 0
        for token in tokens:
            current.append(token)
            decoded = self.decode_with_timestamps(current)
            if replacement_char not in decoded or decoded_full[unicode_offset + decoded.index(replacement_char)] == replacement_char:
                words.append(decoded)
                word_tokens.append(current)
                current = []
                unicode_offset += len(decoded)
        return words, word_tokens

    def split_tokens_on_spaces(self, tokens):
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

    def split_to_word_tokens(self, tokens):
        if self.language in {"zh", "ja", "th", "lo", "my", "yue"}:
            return self.split_tokens_on_unicode(tokens)
        return self.split_tokens_on_spaces(tokens)


def get_tokenizer(multilingual, *, num_languages=99, language=None, task=None, vocab_path=None):
    """whisper.tokenizer.get_tokenizer. Note the reference calls it WITHOUT num_languages
    (infer_ali.py:41), so large-v3 gets the 99-language special-token numbering; kept as is."""
    if language is not None:
        language = language.lower()
        if language not in LANGUAGES:
            if language in TO_LANGUAGE_CODE:
                language = TO_LANGUAGE_CODE[language]
            else:
                raise ValueError("Unsupported language: %s" % language)
    if multilingual:
        language = language or "en"
        task = task or "transcribe"
    else:
        language, task = None, None
    return Tokenizer(multilingual, num_languages, language, task, vocab_path)
