"""Text cleaners (behaviour of neural_speech/utils/text/cleaners.py:45-89).

`convert_to_ascii` stands in for the third-party Unidecode package: ASCII passes through
untouched; other code points go through a punctuation/ligature table, then NFKD decomposition
with combining marks dropped; anything still non-ASCII is removed."""
import re
import unicodedata

from .numbers import normalize_numbers

_whitespace_re = re.compile(r"\s+")

_ABBREVIATIONS = [
    ("mrs", "misess"),   # sic - the reference's spelling, kept for id parity
    ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"),
    ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"),
    ("lt", "lieutenant"), ("hon", "honorable"), ("sgt", "sergeant"), ("capt", "captain"),
    ("esq", "esquire"), ("ltd", "limited"), ("col", "colonel"), ("ft", "fort"),
]
_abbreviations = [(re.compile("\\b%s\\." % a, re.IGNORECASE), b) for a, b in _ABBREVIATIONS]

_TRANSLIT = {
    "‘": "'", "’": "'", "‚": ",", "‛": "'", "“": '"', "”": '"', "„": ",,",
    "‐": "-", "‑": "-", "‒": "-", "–": "-", "—": "--", "―": "--",
    "…": "...", " ": " ", "«": "<<", "»": ">>", "‹": "<", "›": ">",
    "ß": "ss", "æ": "ae", "Æ": "AE", "œ": "oe", "Œ": "OE", "ø": "o",
    "Ø": "O", "đ": "d", "Đ": "D", "ð": "d", "Ð": "D", "þ": "th",
    "Þ": "Th", "ł": "l", "Ł": "L", "ı": "i", "´": "'", "′": "'", "″": '"',
    "×": "x", "÷": "/", "°": "deg", "€": "EUR", "£": "PS", "©": "(c)",
    "®": "(r)", "™": "(tm)", "½": " 1/2", "¼": " 1/4", "¾": " 3/4", "·": "*",
    "•": "*", "¿": "?", "¡": "!",
}


def convert_to_ascii(text):
    if text.isascii():
        return text
    out = []
    for ch in text:
        if ord(ch) < 128:
            out.append(ch)
        elif ch in _TRANSLIT:
            out.append(_TRANSLIT[ch])
        else:
            dec = unicodedata.normalize("NFKD", ch)
            out.append("".join(c for c in dec if ord(c) < 128))
    return "".join(out)


def expand_abbreviations(text):
    for regex, replacement in _abbreviations:
        text = re.sub(regex, replacement, text)
    return text


def expand_numbers(text):
    return normalize_numbers(text)


def lowercase(text):
    return text.lower()


def collapse_whitespace(text):
    return re.sub(_whitespace_re, " ", text)


def basic_cleaners(text):
    return collapse_whitespace(lowercase(text))


def transliteration_cleaners(text):
    return collapse_whitespace(lowercase(convert_to_ascii(text)))


def english_cleaners(text):
    text = convert_to_ascii(text)
    text = lowercase(text)
    text = expand_numbers(text)
    text = expand_abbreviations(text)
    text = collapse_whitespace(text)
    return text
