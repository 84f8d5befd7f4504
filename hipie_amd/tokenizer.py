"""BERT (uncased) WordPiece tokenisation without a `transformers` dependency.

The reference tokenises captions with `AutoTokenizer.from_pretrained("projects/HIPIE/bert-base-uncased")` (hipie_img.py:153,
data/coco_dataset_mapper_uni.py:277) and needs three things from it: the ids / attention mask of a (padded, truncated) batch
(hipie_img.py:904-909), the id of "." (bert_model.py:68-73) and `char_to_token` for the class-prompt map (coco_dataset_mapper_uni.py:
1024-1058).  This is the published BERT algorithm (BasicTokenizer: text cleaning, CJK spacing, accent stripping, lower-casing,
punctuation splitting; WordPiece: greedy longest-match-first with the "##" continuation prefix, [UNK] for unmatched words) with
character offsets carried through every step, reading the same `vocab.txt`.  tests/test_host_logic.py checks ids, masks, offsets and
char_to_token against `transformers.BertTokenizerFast` on the same vocabulary.
"""
import os
import unicodedata

import torch


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class Encoding(object):
    """what the callers read from a HuggingFace BatchEncoding: tensors, `.to(device)`, and char_to_token for ONE sequence."""

    def __init__(self, input_ids, attention_mask, offsets, special_tokens_mask=None):
        self.input_ids, self.attention_mask = input_ids, attention_mask
        self.offsets = offsets                      # per sequence: [(start, end)] per token, (0, 0) for special / padding tokens
        self.special_tokens_mask = special_tokens_mask

    def __getitem__(self, k):
        return getattr(self, k)

    def keys(self):
        return ["input_ids", "attention_mask"] + (["special_tokens_mask"] if self.special_tokens_mask is not None else [])

    def to(self, device):
        self.input_ids, self.attention_mask = self.input_ids.to(device), self.attention_mask.to(device)
        if self.special_tokens_mask is not None:
            self.special_tokens_mask = self.special_tokens_mask.to(device)
        return self

    def char_to_token(self, char_index, batch_index=0):
        if char_index < 0:
            raise IndexError(char_index)
        for t, (s, e) in enumerate(self.offsets[batch_index]):
            if s <= char_index < e:
                return t
        return None


class BertWordPiece(object):
    def __init__(self, vocab_file, do_lower_case=True, unk="[UNK]", cls="[CLS]", sep="[SEP]", pad="[PAD]", max_chars_per_word=100):
        with open(vocab_file, encoding="utf-8") as f:
            toks = [l.rstrip("\n") for l in f]
        self.vocab = {t: i for i, t in enumerate(toks)}
        self.inv = toks
        self.lower = do_lower_case
        self.unk_id, self.cls_id, self.sep_id, self.pad_id = (self.vocab[t] for t in (unk, cls, sep, pad))
        self.max_chars = max_chars_per_word

    @staticmethod
    def from_dir(path):
        """the tokenizer of a HuggingFace model directory (vocab.txt), e.g. projects/HIPIE/bert-base-uncased; None when absent."""
        vf = os.path.join(path, "vocab.txt")
        return BertWordPiece(vf) if os.path.exists(vf) else None

    # ---- BasicTokenizer with offsets ------------------------------------------------------------------------
    def _words(self, text):
        """-> list of words, a word = list of (normalised char, index of the original character it came from)."""
        norm = []
        for i, ch in enumerate(text):
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                norm.append((" ", i))
            elif _is_cjk(cp):
                norm += [(" ", i), (ch, i), (" ", i)]
            elif self.lower:
                for c in unicodedata.normalize("NFD", ch):          # strip accents, then lower-case (BertNormalizer's order)
                    if unicodedata.category(c) != "Mn":
                        norm += [(c2, i) for c2 in c.lower()]
            else:
                norm.append((ch, i))
        words, cur = [], []
        for c, i in norm:
            if c == " ":
                if cur:
                    words.append(cur)
                    cur = []
            elif _is_punctuation(c):
                if cur:
                    words.append(cur)
                    cur = []
                words.append([(c, i)])
            else:
                cur.append((c, i))
        if cur:
            words.append(cur)
        return words

    def _wordpiece(self, word):
        """-> [(id, start, end)] with character offsets into the original text."""
        chars = "".join(c for c, _ in word)
        span = (word[0][1], word[-1][1] + 1)
        if len(chars) > self.max_chars:
            return [(self.unk_id, span[0], span[1])]
        out, start = [], 0
        while start < len(chars):
            end, cur = len(chars), None
            while start < end:
                sub = chars[start:end] if start == 0 else "##" + chars[start:end]
                if sub in self.vocab:
                    cur = self.vocab[sub]
                    break
                end -= 1
            if cur is None:
                return [(self.unk_id, span[0], span[1])]
            out.append((cur, word[start][1], word[end - 1][1] + 1))
            start = end
        return out

    def tokenize_with_offsets(self, text):
        out = []
        for w in self._words(text):
            out += self._wordpiece(w)
        return out

    def convert_ids_to_tokens(self, ids):
        return [self.inv[int(i)] for i in ids]

    # ---- the HuggingFace call surface the reference uses ------------------------------------------------------
    def __call__(self, text, max_length=None, padding=False, truncation=False, return_tensors="pt", return_special_tokens_mask=False,
                 **unused):
        texts = [text] if isinstance(text, str) else list(text)
        rows = []
        for t in texts:
            pieces = self.tokenize_with_offsets(t)
            if truncation and max_length is not None and len(pieces) > max_length - 2:
                pieces = pieces[:max_length - 2]
            rows.append([(self.cls_id, 0, 0)] + pieces + [(self.sep_id, 0, 0)])
        if padding == "max_length" and max_length is not None:
            L = max_length
        elif padding in (True, "longest") or len(rows) > 1:
            L = max(len(r) for r in rows)
        else:
            L = len(rows[0])
        ids = torch.full((len(rows), L), self.pad_id, dtype=torch.long)
        mask = torch.zeros(len(rows), L, dtype=torch.long)
        special = torch.ones(len(rows), L, dtype=torch.long)
        offsets = []
        for b, r in enumerate(rows):
            ids[b, :len(r)] = torch.tensor([p[0] for p in r])
            mask[b, :len(r)] = 1
            special[b, 1:len(r) - 1] = 0
            offsets.append([(s, e) for _, s, e in r] + [(0, 0)] * (L - len(r)))
        return Encoding(ids, mask, offsets, special if return_special_tokens_mask else None)

    batch_encode_plus = __call__
