"""Number normalisation (behaviour of neural_speech/utils/text/numbers.py:43-69).

The reference delegates spelling-out to the third-party `inflect` package (not pinned, not
installable here); `number_to_words` below restates the subset of its behaviour the reference
uses: cardinals with andword='', ordinals, and the year style (group=2, zero='oh')."""
import re

_comma_number_re = re.compile(r"([0-9][0-9,]+[0-9])")
_decimal_number_re = re.compile(r"([0-9]+\.[0-9]+)")
_pounds_re = re.compile(r"£([0-9,]*[0-9]+)")
_dollars_re = re.compile(r"\$([0-9\.,]*[0-9]+)")
_ordinal_re = re.compile(r"[0-9]+(st|nd|rd|th)")
_number_re = re.compile(r"[0-9]+")

_UNITS = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven",
          "twelve", "thirteen", "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_TENS = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
_SCALES = ["", " thousand", " million", " billion", " trillion", " quadrillion", " quintillion"]
_ORD_IRREGULAR = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth",
                  "nine": "ninth", "twelve": "twelfth"}


def _two_digits(n, zero="zero"):
    if n < 20:
        return _UNITS[n] if n else zero
    t, u = divmod(n, 10)
    return _TENS[t] + ("-" + _UNITS[u] if u else "")


def _three_digits(n, andword):
    h, rest = divmod(n, 100)
    parts = []
    if h:
        parts.append(_UNITS[h] + " hundred")
    if rest:
        if h and andword:
            parts.append(andword)
        parts.append(_two_digits(rest))
    return " ".join(parts)


def number_to_words(num, andword="and", zero="zero", group=0):
    """Spell out a non-negative integer the way inflect.engine().number_to_words does."""
    n = int(num)
    if group == 2:
        digits = str(n)
        chunks = [digits[i:i + 2] for i in range(0, len(digits), 2)]
        words = []
        for ch in chunks:
            if len(ch) == 1:
                words.append(_UNITS[int(ch)] if int(ch) else zero)
            elif ch[0] == "0":
                words.append(zero + " " + (_UNITS[int(ch[1])] if int(ch[1]) else zero))
            else:
                words.append(_two_digits(int(ch)))
        return ", ".join(words)
    if n == 0:
        return zero
    groups = []
    while n:
        n, g3 = divmod(n, 1000)
        groups.append(g3)
    out = []
    for i in range(len(groups) - 1, -1, -1):
        if groups[i]:
            # inflect puts the 'and' before a trailing sub-hundred group as well ("one thousand and five")
            if i == 0 and groups[i] < 100 and len(groups) > 1 and andword:
                out.append(andword + " " + _two_digits(groups[i]))
            else:
                out.append(_three_digits(groups[i], andword) + _SCALES[i])
    text = out[0]
    for piece in out[1:]:
        text += (" " if piece.startswith(andword + " ") and andword else ", ") + piece
    return text


def ordinal_words(text):
    """'21st' -> 'twenty-first' (inflect.number_to_words on an ordinal string)."""
    n = int(re.match(r"[0-9]+", text).group(0))
    words = number_to_words(n)
    head, sep, last = words.rpartition("-") if "-" in words.split(" ")[-1] else words.rpartition(" ")
    if last in _ORD_IRREGULAR:
        last = _ORD_IRREGULAR[last]
    elif last.endswith("y"):
        last = last[:-1] + "ieth"
    else:
        last = last + "th"
    return head + sep + last


def _remove_commas(m):
    return m.group(1).replace(",", "")


def _expand_decimal_point(m):
    return m.group(1).replace(".", " point ")


def _expand_dollars(m):
    match = m.group(1)
    parts = match.split(".")
    if len(parts) > 2:
        return match + " dollars"
    dollars = int(parts[0]) if parts[0] else 0
    cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
    if dollars and cents:
        return "%s %s, %s %s" % (dollars, "dollar" if dollars == 1 else "dollars", cents,
                                 "cent" if cents == 1 else "cents")
    if dollars:
        return "%s %s" % (dollars, "dollar" if dollars == 1 else "dollars")
    if cents:
        return "%s %s" % (cents, "cent" if cents == 1 else "cents")
    return "zero dollars"


def _expand_ordinal(m):
    return ordinal_words(m.group(0))


def _expand_number(m):
    num = int(m.group(0))
    if 1000 < num < 3000:
        if num == 2000:
            return "two thousand"
        if 2000 < num < 2010:
            return "two thousand " + number_to_words(num % 100)
        if num % 100 == 0:
            return number_to_words(num // 100) + " hundred"
        return number_to_words(num, andword="", zero="oh", group=2).replace(", ", " ")
    return number_to_words(num, andword="")


def normalize_numbers(text):
    text = re.sub(_comma_number_re, _remove_commas, text)
    text = re.sub(_pounds_re, r"\1 pounds", text)
    text = re.sub(_dollars_re, _expand_dollars, text)
    text = re.sub(_decimal_number_re, _expand_decimal_point, text)
    text = re.sub(_ordinal_re, _expand_ordinal, text)
    text = re.sub(_number_re, _expand_number, text)
    return text
