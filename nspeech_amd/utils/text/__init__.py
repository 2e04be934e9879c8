"""Text front end: string -> cleaned string -> symbol ids (the bit-exact integer path).

Contract (neural_speech/utils/text/__init__.py:14-75): text outside braces runs through the named cleaners and is
mapped character by character; a `{...}` span holds space-separated ARPAbet symbols ('@'-prefixed in the table); the
pad '_' and end-of-sequence '~' symbols never come out of text; id 1 ('~') closes every sequence.

The reference peels `{...}` spans off with the pattern (.*?)\\{(.+?)\\}(.*) applied to the remaining text in a loop;
`spans()` restates that as a scanner, including the corner cases the pattern implies: a span needs at least one
character between its braces (so "{}" is plain text and the search moves on to the next '{'), '.' does not cross a
line break (a newline before the closing brace makes the rest plain text, a newline after it ends the input)."""
from . import cleaners
from .symbols import symbols

SYMBOL_ID = {sym: idx for idx, sym in enumerate(symbols)}
ID_SYMBOL = dict(enumerate(symbols))
_NEVER_FROM_TEXT = ("_", "~")


def spans(text):
    """Yields (plain, arpabet) pieces in order; `arpabet` is None for the trailing plain piece."""
    rest = text
    while rest:
        line_end = rest.find("\n")
        limit = len(rest) if line_end < 0 else line_end          # '.' stops at a line break
        found = None
        start = rest.find("{", 0, limit)
        while start >= 0:
            close = rest.find("}", start + 2, limit)              # at least one character inside
            if close >= 0:
                found = (start, close)
                break
            start = rest.find("{", start + 1, limit)
        if found is None:
            yield rest, None
            return
        start, close = found
        yield rest[:start], rest[start + 1:close]
        rest = rest[close + 1:limit]                               # the pattern's last group ends at the line break


def apply_cleaners(text, cleaner_names):
    for name in cleaner_names:
        fn = getattr(cleaners, name, None)
        if fn is None:
            raise Exception("Unknown cleaner: %s" % name)
        text = fn(text)
    return text


def ids_of(symbol_list):
    return [SYMBOL_ID[s] for s in symbol_list if s in SYMBOL_ID and s not in _NEVER_FROM_TEXT]


def text_to_sequence(text, cleaner_names):
    ids = []
    for plain, arpabet in spans(text):
        ids.extend(ids_of(apply_cleaners(plain, cleaner_names)))
        if arpabet is not None:
            ids.extend(ids_of("@" + s for s in arpabet.split()))
    ids.append(SYMBOL_ID["~"])
    return ids


def sequence_to_text(sequence):
    pieces = []
    for i in sequence:
        sym = ID_SYMBOL.get(int(i))
        if sym is None:
            continue
        pieces.append("{%s}" % sym[1:] if len(sym) > 1 and sym.startswith("@") else sym)
    return "".join(pieces).replace("}{", " ")


# the reference's private helper names, for callers that reach for them
_symbol_to_id, _id_to_symbol = SYMBOL_ID, ID_SYMBOL
_clean_text = apply_cleaners
_symbols_to_sequence = ids_of


def _arpabet_to_sequence(text):
    return ids_of("@" + s for s in text.split())


def _should_keep_symbol(s):
    return s in SYMBOL_ID and s not in _NEVER_FROM_TEXT
