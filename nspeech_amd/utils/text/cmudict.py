"""CMU pronouncing dictionary reader with the surface of neural_speech/utils/text/cmudict.py:16-60:
`valid_symbols` (84 phones: 24 consonants, 15 vowels bare and with stress 0/1/2, in alphabetical order of the base
phone), `CMUDict(path_or_file, keep_ambiguous=True)`, `len()`, `.lookup(word)` -> list of pronunciations or None."""
import string

VOWEL_PHONES = "AA AE AH AO AW AY EH ER EY IH IY OW OY UH UW".split()
CONSONANT_PHONES = "B CH D DH F G HH JH K L M N NG P R S SH T TH V W Y Z ZH".split()

valid_symbols = [phone + stress
                 for phone in sorted(VOWEL_PHONES + CONSONANT_PHONES)
                 for stress in ([""] + list("012") if phone in VOWEL_PHONES else [""])]
_PHONE_SET = frozenset(valid_symbols)
_WORD_START = frozenset(string.ascii_uppercase + "'")


def _strip_variant(word):
    """'READ(2)' -> 'READ': alternative pronunciations carry a parenthesised counter."""
    out, i = [], 0
    while i < len(word):
        if word[i] == "(":
            j = i + 1
            while j < len(word) and word[j] in "0123456789":
                j += 1
            if j > i + 1 and j < len(word) and word[j] == ")":      # every '(digits)' group goes, wherever it sits
                i = j + 1
                continue
        out.append(word[i])
        i += 1
    return "".join(out)


def _entries(lines):
    """(word, 'P1 P2 ...') for every dictionary line whose phones are all known; comment lines start with ';;;'."""
    for raw in lines:
        if not raw or raw[0] not in _WORD_START:
            continue
        fields = raw.split("  ")
        phones = fields[1].strip().split(" ")
        if all(p in _PHONE_SET for p in phones):
            yield _strip_variant(fields[0]), " ".join(phones)


class CMUDict(object):
    def __init__(self, file_or_path, keep_ambiguous=True):
        if isinstance(file_or_path, str):
            with open(file_or_path, encoding="latin-1") as handle:
                pairs = list(_entries(handle))
        else:
            pairs = list(_entries(file_or_path))
        table = {}
        for word, pron in pairs:
            table.setdefault(word, []).append(pron)
        self._table = table if keep_ambiguous else {w: p for w, p in table.items() if len(p) == 1}

    def __len__(self):
        return len(self._table)

    def lookup(self, word):
        return self._table.get(word.upper())
