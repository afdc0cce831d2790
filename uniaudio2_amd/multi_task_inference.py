#!/usr/bin/env python3
"""Command line of the hot path — mirror of the reference's multi_task_inference.py (flags :554-598,
validation :601-649, run_understanding :258-382, run_generation_stage1 :408-525, stage 2 :529-549) on the
MI355X kernels.  Same flags, same `{name}_reason.pt` / `{name}_semantic.pt` (8, T) tensor contract
(generation writes int32, :523-524), same error messages for missing inputs.

    python -m uniaudio2_amd.multi_task_inference --task TTS --stage 1 --text "..." --topk 1 \\
        --llm_train_config cfg.yaml --resume llm.checkpoint --text_tokenizer_path tok_dir --prompt_json prompts.json

Differences, all explicit:
  * `--dtype {bf16,fp32}` (default bf16) picks the kernel precision; the reference runs fp32.
  * `--batch_size N` decodes N utterances of a --text_file together per GPU (continuous batching; TTS).
  * Under `torchrun` (WORLD_SIZE > 1) the texts of --text_file are sharded over the ranks, one utterance
    per GPU at a time, and gathered with one RCCL all-gather (uniaudio2_amd/parallel.py); rank 0 writes.
  * --topk > 1 samples on the device with a counter-based generator seeded by --seed: reproducible, same
    distribution as the reference's sample_topk, but not torch's random stream.  --decode_type
    ngram/beamsearch raise NotImplementedError (the reference's beam search is dead code).
  * Encoding raw audio (--audio / --audio_dir) needs the codec's frozen SSL encoders (Whisper, WavLM, BEST-RQ), which are
    out of scope: it raises with that message; pre-tokenised `--reason_pt/--semantic_pt/--token_dir`
    inputs work.  Stage 2 (tokens -> wav: RVQ look-ups, flow-matching DiT, SQ-Codec) runs on the device; wav files are
    written with scipy (16-bit PCM) because torchaudio is not a dependency.
"""
import argparse
import glob
import json
import os
import random

import torch
import yaml

from .llm_models.model_new import Model_stage3, ModelArgs

UNDERSTANDING_TASKS = ["ASR", "Yue_ASR", "lyric_recognition", "audio_caption", "music_caption", "audio_understanding", "speech_s2t"]
GENERATION_TASKS = ["TTS", "Yue_TTS", "TTA", "TTM", "LTS", "InstructTTS", "speech_s2s"]
TASK_PROMPT_SUFFIX = "\n\n"


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def _prompt_key_from_task(task):
    t = task.strip().lower()
    fixed = {"yue_tts": "Yue_TTS", "yue_asr": "Yue_ASR", "instruct_tts": "InstructTTS", "speech_s2s": "speech_s2s",
             "speech_s2t": "speech_s2t"}
    if t in fixed:
        return fixed[t]
    return t.upper() if t in ("asr", "tts", "tta", "ttm", "lts") else t


def _get_prompt_tensor(args, text_tokenizer, task_name):
    prompt_text = (getattr(args, "prompt_text", None) or "").strip()
    path = getattr(args, "prompt_json", None)
    if prompt_text:
        chosen = prompt_text
    elif path and os.path.isfile(path):
        with open(path, "r", encoding="utf-8") as f:
            prompts = json.load(f)
        key = _prompt_key_from_task(task_name)
        if key not in prompts:
            key = task_name if task_name in prompts else task_name.upper()
        if key not in prompts:
            key = list(prompts.keys())[0]
        if not prompts[key]:
            raise ValueError(f"Task '{key}' has no prompts in {path}.")
        chosen = random.choice(prompts[key])
    else:
        raise ValueError("Provide --prompt_text or --prompt_json.")
    return torch.tensor(text_tokenizer.tokenize(chosen.strip() + TASK_PROMPT_SUFFIX), dtype=torch.long)


def resume_for_inference(resume, exp_dir, model, device):
    """llm_utils/train_utils.py:159-177: {'model': state_dict}, 'module.' prefixes stripped; latest epN.checkpoint of exp_dir otherwise."""
    if resume is None and exp_dir:
        cands = sorted(glob.glob(os.path.join(exp_dir, "ep*.checkpoint")), key=os.path.getmtime)
        resume = cands[-1] if cands else None
    if resume is None:
        raise ValueError("Set --resume or --exp_dir with a checkpoint.")
    ckpt = torch.load(resume, map_location="cpu")
    sd = ckpt["model"] if "model" in ckpt else ckpt
    sd = {(k.split("module.")[-1] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd)
    return model


def _load_config_and_llm(args):
    with open(args.llm_train_config, "r", encoding="utf-8") as f:
        train_args = argparse.Namespace(**yaml.safe_load(f))
    torch.manual_seed(args.seed)
    random.seed(args.seed)            # every rank draws the same task prompt from --prompt_json (random.choice below)
    if not torch.cuda.is_available():
        raise RuntimeError("uniaudio2_amd needs a ROCm GPU (no CPU fallback)")
    rank = int(os.environ.get("LOCAL_RANK", getattr(args, "rank", 0)))
    device = torch.device(f"cuda:{rank % torch.cuda.device_count()}")
    torch.cuda.set_device(device)
    config = ModelArgs(decoder_name=train_args.local_model, llm_pretrained_model=train_args.llm_pretrained_model,
                       llm_name=train_args.llm_name, audio_semantic_vocab_size=train_args.audio_semantic_card,
                       audio_reason_vocab_size=train_args.audio_reason_card, audio_num_codebooks=train_args.parallel_number - 1,
                       audio_embeddings_path=train_args.audio_embeddings_path,
                       audio_understanding_expert_path=train_args.audio_understanding_expert_path)
    model = Model_stage3(config)
    resume_for_inference(args.resume, args.exp_dir, model, device)
    model.to(device=device, dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model.order_free_rows = max(0, int(getattr(args, "order_free_rows", 0) or 0))     # applied by setup_caches (bf16 plans)
    return train_args, model, device


def _get_generator_class(task):
    task = task.strip().lower()
    from .evaluation import (asr_task, audio_music_caption_task, audio_understanding, audiogen_task, insturct_tts_task, lyric_asr_task,
                             musicgen_task, songen_task, speech_s2s, speech_s2t, tts_task)
    # the reference's routing, module for module (multi_task_inference.py:187-255)
    table = {"asr": asr_task, "yue_asr": asr_task, "lyric_recognition": lyric_asr_task, "audio_caption": audio_music_caption_task,
             "music_caption": audio_music_caption_task, "audio_understanding": audio_understanding, "speech_s2t": speech_s2t,
             "tts": tts_task, "yue_tts": tts_task, "tta": audiogen_task, "ttm": musicgen_task, "lts": songen_task,
             "instruct_tts": insturct_tts_task, "instructtts": insturct_tts_task, "speech_s2s": speech_s2s}
    if task not in table:
        raise ValueError(f"Unknown task: {task}. Understanding: {UNDERSTANDING_TASKS}. Generation: {GENERATION_TASKS}.")
    return table[task].Generator


def _save_tokens(args, name, reason, semantic):
    """The reference's on-disk contract (multi_task_inference.py:522-523): `{name}_reason.pt` / `{name}_semantic.pt`, each a
    torch.save of an (8, T) int32 CPU tensor.  `--save_safetensors` adds `{name}_tokens.safetensors` with the same two
    tensors (keys "reason", "semantic") for consumers that do not unpickle."""
    reason, semantic = reason.cpu().contiguous(), semantic.cpu().contiguous()
    torch.save(reason, os.path.join(args.output_dir, f"{name}_reason.pt"))
    torch.save(semantic, os.path.join(args.output_dir, f"{name}_semantic.pt"))
    if getattr(args, "save_safetensors", False):
        from safetensors.torch import save_file
        save_file({"reason": reason, "semantic": semantic}, os.path.join(args.output_dir, f"{name}_tokens.safetensors"))


def _generation_method_name(task):
    t = task.strip().lower()
    if t in ("tts", "yue_tts"):
        return "generate_tts"
    if t in ("tta", "ttm"):
        return "generate_audio"
    if t == "lts":
        return "generate_LTS"
    if t in ("instruct_tts", "instructtts"):
        return "generate_instruct_tts"
    if t == "speech_s2s":
        return "generate_audio"
    raise ValueError(f"Unknown generation task: {task}")


def _check_decode(args):
    if args.decode_type != "greedy":
        raise NotImplementedError("--decode_type ngram/beamsearch are not on the device path (greedy only this round)")


def run_understanding(args):
    task = args.task.strip().lower()
    _check_decode(args)
    if (args.audio and os.path.isfile(args.audio)) or (args.audio_dir and os.path.isdir(args.audio_dir)):
        raise NotImplementedError("encoding raw audio needs the codec's frozen SSL encoders (out of scope, SURVEY.md §8f); "
                                  "pass --reason_pt + --semantic_pt or --token_dir")
    if args.reason_pt and args.semantic_pt and os.path.isfile(args.reason_pt) and os.path.isfile(args.semantic_pt):
        token_dir = os.path.dirname(args.reason_pt) or "."
        names = [os.path.basename(args.reason_pt).replace("_reason.pt", "")]
    elif args.token_dir and os.path.isdir(args.token_dir):
        names = [os.path.basename(p).replace("_reason.pt", "") for p in sorted(glob.glob(os.path.join(args.token_dir, "*_reason.pt")))]
        token_dir = args.token_dir
    else:
        raise ValueError("For understanding task provide --audio/--audio_dir, --reason_pt+--semantic_pt, or --token_dir with *_reason.pt.")
    train_args, model, device = _load_config_and_llm(args)
    generator = _get_generator_class(task)(model, train_args, audio_tokenizer_config=args.audio_tokenizer_config,
                                           audio_model_path=args.audio_model_path, text_tokenizer_path=args.text_tokenizer_path,
                                           is_cfg=args.use_cfg)
    task_prompt = _get_prompt_tensor(args, generator._text_tokenizer, args.task)
    results_path = args.results or os.path.join(args.output_dir, f"{task}_results.txt")
    os.makedirs(os.path.dirname(results_path) or ".", exist_ok=True)
    with open(results_path, "w") as f_out:
        for name in names:
            rp, sp = os.path.join(token_dir, f"{name}_reason.pt"), os.path.join(token_dir, f"{name}_semantic.pt")
            if not os.path.isfile(rp) or not os.path.isfile(sp):
                print(f"[Skip] {name}: missing reason/semantic .pt")
                continue
            reason = torch.load(rp, map_location="cpu").transpose(0, 1).long()          # (T, 8), :304-308
            semantic = torch.load(sp, map_location="cpu").transpose(0, 1).long()
            if task == "audio_understanding":
                question = (args.question or "").strip()
                if not question and args.question_file and os.path.isfile(args.question_file):      # :345-347
                    with open(args.question_file, "r", encoding="utf-8") as f:
                        question = f.read().strip()
                question = question or "What is described in this audio?"
                d = {"text_seq_question": torch.tensor(generator._text_tokenizer.tokenize(question), dtype=torch.long),
                     "reason_seq": reason.transpose(0, 1), "semantic_seq": semantic.transpose(0, 1)}
                text_out = generator.generate_answer(task_prompt, task_name=task, d=d, keys=list(d), types=["text", "audio", "audio"],
                                                     temperature=args.temperature, topk=1, cfg_scale=args.cfg_scale)
            elif task == "speech_s2t":                                                             # :363-379: samples with --topk
                d = {"reason_seq": reason, "semantic_seq": semantic}                                # (T, 8) as the reference passes them (:368)
                result = generator.generate_answer(task_prompt, task_name="speech_s2t", d=d, keys=["reason_seq", "semantic_seq"],
                                                   types=["audio", "audio"], temperature=args.temperature, topk=args.topk,
                                                   cfg_scale=args.cfg_scale)
                if result == (-1, -1):                                                             # :379-381
                    print(f"[Skip] {name}: sequence too long for speech_s2t")
                    continue
                text_out = result[0] if isinstance(result, tuple) else result
            elif task in ("audio_caption", "music_caption"):                                       # :329-342
                text_out = generator.generate_audio_caption(task_prompt, task_name=task, reason_token=reason, semantic_token=semantic,
                                                            temperature=args.temperature, topk=1, cfg_scale=args.cfg_scale)
            else:   # asr, yue_asr, lyric_recognition (:310-327: the reference calls generate_asr on all three; its lyric module only
                    # defines generate_lyric_asr, so the call fails there — here the lyric Generator has both names)
                text_out = generator.generate_asr(task_prompt, task_name=task, reason_token=reason, semantic_token=semantic,
                                                  temperature=args.temperature, topk=1, cfg_scale=args.cfg_scale)
            f_out.write(f"{name}\t{text_out}\n")
            print(f"[{task}] {name} -> {text_out[:80]}...")
    print(f"Results written to {results_path}")


def run_generation_stage1(args):
    import torch.distributed as dist
    from . import parallel
    task = args.task.strip().lower()
    _check_decode(args)
    os.makedirs(args.output_dir, exist_ok=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl")
    train_args, model, device = _load_config_and_llm(args)
    generator = _get_generator_class(task)(model, train_args, audio_tokenizer_config=args.audio_tokenizer_config,
                                           audio_model_path=args.audio_model_path, text_tokenizer_path=args.text_tokenizer_path,
                                           is_cfg=args.use_cfg)
    tok = generator._text_tokenizer
    task_prompt = _get_prompt_tensor(args, tok, args.task)
    if task == "speech_s2s":
        return _run_speech_s2s(args, generator, task_prompt)
    if args.text and args.text.strip():
        items = [("utt_0", args.text.strip())]
    elif args.text_file and os.path.isfile(args.text_file):
        with open(args.text_file, "r", encoding="utf-8") as f:
            items = [(f"utt_{i}", line.strip()) for i, line in enumerate(f) if line.strip()]
    else:
        raise ValueError("Generation requires --text or --text_file.")
    if not items:
        raise ValueError("No text input.")
    gen_fn = getattr(generator, _generation_method_name(task))
    ids = [torch.tensor(tok.tokenize(t), dtype=torch.long) for _, t in items]

    extra = (lambda i: {"caption": ids[i]}) if _generation_method_name(task) == "generate_instruct_tts" else (lambda i: {})   # :520-521

    def one(i):
        # the sampler's key folds in the global utterance index and a new key rewinds the device's draw index
        # (ua2_stage3_set_sampling), so with one utterance at a time an utterance's samples depend only on (seed, index):
        # not on the world size, nor on what its rank generated before it (tests/test_gpu_lm.py)
        model.sampling_seed = parallel.utterance_seed(args.seed, i)
        return gen_fn(task_prompt=task_prompt, task_name=task, text_token=ids[i], temperature=args.temperature,
                      topk=args.topk, cfg_scale=args.cfg_scale, **extra(i))

    if args.batch_size > 1 and hasattr(generator, "generate_tts_batch") and _generation_method_name(task) == "generate_tts":
        def many(idx):
            # batched chunks are keyed by their first utterance and draw per (row, frame): samples here DO depend on the
            # chunk composition, i.e. on --batch_size and the world size (greedy decoding, topk = 1, does not)
            model.sampling_seed = parallel.utterance_seed(args.seed, idx[0])
            return generator.generate_tts_batch(task_prompt=task_prompt, task_name=task, text_tokens=[ids[i] for i in idx],
                                                temperature=args.temperature, topk=args.topk, cfg_scale=args.cfg_scale)
        results = parallel.run_sharded_batched(list(range(len(items))), [len(x) for x in ids], many, args.batch_size)
    else:
        results = parallel.run_sharded(list(range(len(items))), [len(x) for x in ids], one)
    _write_results(args, items, results, "[Stage1]")
    return args.output_dir


def _write_results(args, items, results, tag):
    """Rank 0 writes the gathered token tensors; an utterance whose generation failed on its rank (parallel.Failed:
    e.g. the model never produced the semantic phase) is reported after everything else was saved."""
    from . import parallel
    failed = []
    rank0 = int(os.environ.get("RANK", "0")) == 0
    for i, (name, _) in enumerate(items):                       # every rank holds every result (the all-gather): all of them raise
        if isinstance(results[i], parallel.Failed):             # together, none is left waiting in stage 2's barrier
            failed.append(name)
            if rank0:
                print(f"[Fail] {name}: {results[i].message}")
            continue
        if rank0:
            reason, semantic = results[i]
            _save_tokens(args, name, reason, semantic)
            print(f"{tag} {name} -> {name}_reason.pt, {name}_semantic.pt")
    if failed:
        raise RuntimeError(f"{len(failed)} of {len(items)} utterances failed: {', '.join(failed)}")


def _run_speech_s2s(args, generator, task_prompt):
    """multi_task_inference.py:414-483: source = *_reason.pt / *_semantic.pt pairs of --token_dir; output = generated pairs."""
    import glob
    from . import parallel
    from .evaluation.speech_s2s import S2S_KEYS, S2S_TYPES
    if (args.audio and os.path.isfile(args.audio)) or (args.audio_dir and os.path.isdir(args.audio_dir)):
        raise NotImplementedError("encoding raw audio needs the codec's frozen SSL encoders (out of scope, SURVEY.md §8f); "
                                  "pass --token_dir with the source *_reason.pt / *_semantic.pt")
    if not (args.token_dir and os.path.isdir(args.token_dir)):
        raise ValueError("speech_s2s requires --audio, --audio_dir, or --token_dir (source reason/semantic .pt).")
    names = [os.path.basename(p).replace("_reason.pt", "") for p in sorted(glob.glob(os.path.join(args.token_dir, "*_reason.pt")))]
    items = []
    for name in names:
        rp, sp = os.path.join(args.token_dir, f"{name}_reason.pt"), os.path.join(args.token_dir, f"{name}_semantic.pt")
        if not os.path.isfile(sp):
            print(f"[Skip] {name}: missing source {rp} or {sp}")
            continue
        reason, semantic = torch.load(rp, map_location="cpu"), torch.load(sp, map_location="cpu")
        items.append((name, {"reason_seq_1": reason, "semantic_seq_1": semantic, "reason_seq_2": reason, "semantic_seq_2": semantic}))

    def one(i):
        generator._model.sampling_seed = parallel.utterance_seed(args.seed, i)
        return generator.generate_audio(task_prompt=task_prompt, task_name="speech_s2s", d=items[i][1], keys=S2S_KEYS[:-2],
                                        types=S2S_TYPES[:-2], temperature=args.temperature, topk=args.topk, cfg_scale=args.cfg_scale)

    results = parallel.run_sharded(list(range(len(items))), [int(d["semantic_seq_1"].shape[-1]) for _, d in items], one)
    _write_results(args, items, results, "[Stage1] speech_s2s")
    return args.output_dir


def _load_codec(args, device):
    """multi_task_inference.py:84-97."""
    from .tools.tokenizer.ReasoningCodec_film.reason_tokenizer import ReasoningTokenizer
    if not args.codec_config:
        raise ValueError("Stage 2 requires --codec_config (the codec's infer yaml) and --codec_ckpt.")
    return ReasoningTokenizer(train_config=args.codec_config, model_path=args.codec_ckpt, music_ssl_folder=args.music_ssl_folder, device=device)


def save_wav(path, wave, sample_rate):
    """torchaudio.save(path, wave (1, N) float32, sr) of the reference (:546) without torchaudio: 16-bit PCM, the
    default encoding torchaudio picks for float input to a .wav."""
    import numpy as np
    from scipy.io import wavfile
    x = wave.detach().cpu().float().clamp(-1.0, 1.0).numpy()
    wavfile.write(path, int(sample_rate), np.round(x.T * 32767.0).astype(np.int16))


def stage2_shard(names, world, rank, lengths=None):
    """Utterances of stage 2 this rank decodes ("codec stage 2 shards the same way ... by utterance"; the reference
    loops over all of them on one GPU, multi_task_inference.py:540-548).  With `lengths` (semantic frames per utterance) the split is
    stage 1's — longest first, dealt round-robin (parallel.shard_indices) — so every rank gets the same mix of 1-, 2- and 3-window
    utterances and a rank's list comes back longest first: consecutive --codec_batch groups then hold utterances with the same
    window count and do not shrink window by window.  Without lengths: names[rank::world] of the sorted list.  The outputs are
    files, so the stage needs no collective: the union over ranks is every name, the shards are disjoint."""
    if lengths is None:
        return list(names[rank::max(1, world)])
    from . import parallel
    return [names[i] for i in parallel.shard_indices(list(lengths), max(1, world), rank)]


def _semantic_frames(path):
    """T of a `{name}_semantic.pt` (8, T) file; 0 when it is missing (the utterance is then skipped with a message)."""
    if not os.path.isfile(path):
        return 0
    return int(torch.load(path, map_location="cpu").shape[-1])


def run_generation_stage2(args):
    """multi_task_inference.py:529-549: every `*_semantic.pt` of --token_dir (default: --output_dir) -> `{wav_dir}/{name}.wav`
    through ReasoningTokenizer.detokenize_no_reason (RVQ look-ups -> flow-matching DiT + guided Euler ODE -> SQ-Codec decode ->
    cross-faded 20-s windows).  Under torchrun every rank decodes its shard of the utterances (stage2_shard), --codec_batch of them
    at a time (detokenize_no_reason_batch: window k of the batch in one DiT solve)."""
    if not torch.cuda.is_available():
        raise RuntimeError("uniaudio2_amd needs a ROCm GPU (no CPU fallback)")
    rank = int(os.environ.get("LOCAL_RANK", getattr(args, "rank", 0)))
    device = torch.device(f"cuda:{rank % torch.cuda.device_count()}")
    torch.cuda.set_device(device)
    return decode_token_dir(_load_codec(args, device), args, device)


def decode_token_dir(codec, args, device):
    """Stage 2 proper, on a loaded codec (anything with detokenize_no_reason / detokenize_no_reason_batch / sample_rate): this
    rank's shard of the token files -> wav files.  RANK / WORLD_SIZE as torchrun sets them."""
    token_dir = args.token_dir or args.output_dir
    names = [os.path.basename(p).replace("_reason.pt", "") for p in sorted(glob.glob(os.path.join(token_dir, "*_reason.pt")))]
    wav_dir = args.wav_dir or os.path.join(token_dir, "wavs")
    os.makedirs(wav_dir, exist_ok=True)
    world, grank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    frames = [_semantic_frames(os.path.join(token_dir, f"{n}_semantic.pt")) for n in names]       # every rank reads the same few bytes
    mine = []
    for name in stage2_shard(names, world, grank, lengths=frames):
        sp = os.path.join(token_dir, f"{name}_semantic.pt")
        if not os.path.isfile(sp):
            print(f"[Skip] {name}: missing {sp}")
            continue
        mine.append((name, sp))
    bs = max(1, int(getattr(args, "codec_batch", 1) or 1))
    for b0 in range(0, len(mine), bs):
        chunk = mine[b0:b0 + bs]
        codes = [torch.load(sp, map_location=device).long() for _, sp in chunk]
        if len(chunk) == 1:
            waves = [codec.detokenize_no_reason(codes[0], return_reasoning_text=False, steps=args.codec_steps)]
        else:
            waves = codec.detokenize_no_reason_batch(codes, steps=args.codec_steps, max_batch=bs)
        for (name, _), wave in zip(chunk, waves):
            wav_path = os.path.join(wav_dir, f"{name}.wav")
            save_wav(wav_path, wave, codec.sample_rate)
            print(f"[Stage2] {name} -> {wav_path}")
    return wav_dir


def get_parser():
    p = argparse.ArgumentParser(description="Multi-task inference: understanding (audio->text) or generation (text->wav)")
    p.add_argument("--task", type=str, required=True)
    p.add_argument("--stage", type=str, default="all", choices=["1", "2", "all"])
    for name in ("audio", "audio_dir", "reason_pt", "semantic_pt", "question", "question_file", "text_file", "results", "token_dir",
                 "wav_dir", "prompt_text", "prompt_json", "llm_train_config", "resume", "exp_dir", "text_tokenizer_path",
                 "audio_tokenizer_config", "audio_model_path", "codec_config", "codec_ckpt", "music_ssl_folder"):
        p.add_argument(f"--{name}", type=str, default=None)
    p.add_argument("--text", type=str, default="")
    p.add_argument("--output_dir", type=str, default="./multi_task_out")
    p.add_argument("--use_cfg", type=str2bool, default=False)
    p.add_argument("--temperature", type=float, default=0.9)
    p.add_argument("--topk", type=int, default=50)
    p.add_argument("--cfg_scale", type=float, default=1.0)
    p.add_argument("--decode_type", type=str, default="greedy", choices=["greedy", "ngram", "beamsearch"])
    p.add_argument("--codec_steps", type=int, default=50)
    p.add_argument("--codec_duration", type=float, default=30.0)
    p.add_argument("--seed", type=int, default=888)
    p.add_argument("--rank", type=int, default=0)
    p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"], help="kernel precision (extension; the reference runs fp32)")
    p.add_argument("--save_safetensors", action="store_true",
                   help="also write {name}_tokens.safetensors next to the reference's two .pt files (extension)")
    p.add_argument("--codec_batch", type=int, default=1,
                   help="utterances whose k-th windows share one DiT solve / SQ-Codec decode in stage 2 (extension; default 1 = one by one as "
                        "the reference: a waveform then never depends on which other files are in the directory, on WORLD_SIZE or on this "
                        "flag.  8 halves the time per window; under the DiT's default order-free GEMM plan a waveform then differs from its "
                        "one-by-one value by bf16 rounding noise (~3e-3 relative rms); UA2_DIT_SUM_ORDER=0 keeps the bits at any batch)")
    p.add_argument("--order_free_rows", type=int, default=0,
                   help="stage 1, bf16 (extension): LM launches of at least this many rows (batched prefill, decode frames of that many "
                        "sequences) take the order-free 256-row-tile GEMM — faster, but a sequence's ids may then depend on what shares "
                        "its batch.  Default 0 = off: every row keeps the bits of its single-sequence run")
    p.add_argument("--batch_size", type=int, default=1,
                   help="utterances decoded together per GPU (extension; TTS / Yue_TTS with --text_file; 1 = one by one as the reference)")
    return p


def main(argv=None):
    args = get_parser().parse_args(argv)
    task = args.task.strip().lower()
    if task in [t.lower() for t in UNDERSTANDING_TASKS]:
        has_input = ((args.audio and os.path.isfile(args.audio)) or (args.audio_dir and os.path.isdir(args.audio_dir)) or
                     (args.reason_pt and args.semantic_pt and os.path.isfile(args.reason_pt) and os.path.isfile(args.semantic_pt)) or
                     (args.token_dir and os.path.isdir(args.token_dir)))
        if not has_input:
            raise ValueError("For understanding task provide --audio, --audio_dir, or --reason_pt + --semantic_pt.")
        if not args.llm_train_config or not args.text_tokenizer_path:
            raise ValueError("Set --llm_train_config and --text_tokenizer_path.")
        if not (args.prompt_text or (args.prompt_json and os.path.isfile(args.prompt_json))):
            raise ValueError("Set --prompt_text or --prompt_json.")
        run_understanding(args)
        return
    if task in [t.lower() for t in GENERATION_TASKS]:
        if task == "speech_s2s":
            if not ((args.audio and os.path.isfile(args.audio)) or (args.audio_dir and os.path.isdir(args.audio_dir)) or
                    (args.token_dir and os.path.isdir(args.token_dir))):
                raise ValueError("speech_s2s requires --audio, --audio_dir, or --token_dir (source reason/semantic .pt).")
        elif not ((args.text and args.text.strip()) or (args.text_file and os.path.isfile(args.text_file))):
            raise ValueError("For generation task provide --text or --text_file.")
        if not args.llm_train_config or not args.text_tokenizer_path:
            raise ValueError("Set --llm_train_config and --text_tokenizer_path.")
        if not (args.prompt_text or (args.prompt_json and os.path.isfile(args.prompt_json))):
            raise ValueError("Set --prompt_text or --prompt_json.")
        if args.stage in ("1", "all"):
            run_generation_stage1(args)
            if args.stage == "1":
                print("[Done] Stage 1 only. Run with --stage 2 --token_dir ... to decode to wav.")
                return
        # stage 2 decodes the token files rank 0 wrote, every rank its shard of the utterances (stage2_shard): the files must be
        # complete before any rank lists them
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()
        run_generation_stage2(args)
        return
    raise ValueError(f"Unsupported task: {task}. Understanding: {UNDERSTANDING_TASKS}. Generation: {GENERATION_TASKS}.")


if __name__ == "__main__":
    main()
