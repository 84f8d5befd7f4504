"""Class-prompt construction for the detection task: category names -> caption + positive_map_label_to_token.

Mirrors create_queries_and_maps / create_positive_dict / clean_name of the reference's test-time mapper
(projects/HIPIE/hipie/data/coco_dataset_mapper_uni.py:54-90, 1024-1058, 732-736): labels are joined with ". ", each label's
character span is mapped to its token span with the tokenizer's char_to_token (with the reference's +-1/2/3 character
fallbacks), and label j (1-based) -> list of token positions.  Pure host-side string work; any HuggingFace *fast* tokenizer
(BatchEncoding.char_to_token) works -- HIPIE uses bert-base-uncased.
"""
import re


def clean_name(name):
    name = re.sub(r"\(.*\)", "", name)
    name = re.sub(r"_", " ", name)
    name = re.sub(r"  ", " ", name)
    return name


def create_positive_dict(tokenized, tokens_positive, labels):
    """token position -> label, and label -> [token positions] (labels start at 1)."""
    token_to_label, label_to_token = {}, {}

    def first_hit(positions):
        for c in positions:
            try:
                t = tokenized.char_to_token(c)
            except Exception:            # the reference swallows out-of-range lookups (:1036-1048)
                return None
            if t is not None:
                return t
        return None
    for j, spans in enumerate(tokens_positive):
        for beg, end in spans:
            beg_pos = first_hit((beg, beg + 1, beg + 2))
            end_pos = first_hit((end - 1, end - 2, end - 3))
            if beg_pos is None or end_pos is None:
                continue
            label_to_token[labels[j]] = list(range(beg_pos, end_pos + 1))
            for i in range(beg_pos, end_pos + 1):
                token_to_label[i] = labels[j]
    return token_to_label, label_to_token


def create_queries_and_maps(categories, tokenizer, things_only=False):
    """categories: [{"name": str, "isthing": 0|1 (optional)}] -> (caption, positive_map_label_to_token)."""
    names = [clean_name(c["name"]) for c in categories if (c.get("isthing", 1) or not things_only)]
    labels = list(range(1, len(names) + 1))
    caption, spans = "", []
    for i, name in enumerate(names):
        start = len(caption)
        caption += name
        spans.append([(start, len(caption))])
        if i != len(names) - 1:
            caption += ". "
    tokenized = tokenizer(caption, return_tensors="pt")
    _, label_to_token = create_positive_dict(tokenized, spans, labels)
    return caption, label_to_token


def detection_inputs(image, categories, tokenizer, is_thing=None):
    """one element of ``batched_inputs`` for HIPIE_IMG.forward (task "detection") from an image tensor and category names."""
    caption, pmap = create_queries_and_maps(categories, tokenizer)
    tok = tokenizer(caption, return_tensors="pt")
    if is_thing is None:
        is_thing = {i + 1: bool(c.get("isthing", 1)) for i, c in enumerate(categories)}
    return {"image": image, "task": "detection", "expressions": caption, "input_ids": tok["input_ids"][0],
            "attention_mask": tok["attention_mask"][0], "positive_map_label_to_token": pmap, "is_thing": is_thing}
