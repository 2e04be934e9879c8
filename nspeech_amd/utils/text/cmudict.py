"""CMUDict reader (behaviour of neural_speech/utils/text/cmudict.py:16-60)."""
import re

_VOWELS = ["AA", "AE", "AH", "AO", "AW", "AY", "EH", "ER", "EY", "IH", "IY", "OW", "OY", "UH", "UW"]
_CONSONANTS = ["B", "CH", "D", "DH", "F", "G", "HH", "JH", "K", "L", "M", "N", "NG", "P", "R", "S", "SH", "T",
               "TH", "V", "W", "Y", "Z", "ZH"]


def _build_symbols():
    # alphabetical order of the base phones; each vowel comes bare and with stress digits 0..2
    out = []
    for ph in sorted(_VOWELS + _CONSONANTS):
        out.append(ph)
        if ph in _VOWELS:
            out.extend(ph + d for d in "012")
    return out


valid_symbols = _build_symbols()
_valid_symbol_set = set(valid_symbols)
_alt_re = re.compile(r"\([0-9]+\)")


class CMUDict(object):
    """word -> list of ARPAbet pronunciations; accepts a path (latin-1) or an open file."""

    def __init__(self, file_or_path, keep_ambiguous=True):
        if isinstance(file_or_path, str):
            with open(file_or_path, encoding="latin-1") as f:
                entries = _parse_cmudict(f)
        else:
            entries = _parse_cmudict(file_or_path)
        if not keep_ambiguous:
            entries = {w: p for w, p in entries.items() if len(p) == 1}
        self._entries = entries

    def __len__(self):
        return len(self._entries)

    def lookup(self, word):
        return self._entries.get(word.upper())


def _parse_cmudict(lines):
    table = {}
    for line in lines:
        if len(line) and ("A" <= line[0] <= "Z" or line[0] == "'"):
            parts = line.split("  ")
            word = re.sub(_alt_re, "", parts[0])
            pron = _get_pronunciation(parts[1])
            if pron:
                table.setdefault(word, []).append(pron)
    return table


def _get_pronunciation(s):
    parts = s.strip().split(" ")
    for part in parts:
        if part not in _valid_symbol_set:
            return None
    return " ".join(parts)
