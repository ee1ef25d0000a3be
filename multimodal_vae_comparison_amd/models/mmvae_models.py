"""The multimodal mixers behind the reference's registry names (reference: models/mmvae_models.py).

`objective()` is the per-step hot path: towers -> ONE fused latent kernel (product of experts, reparameterised
samples, analytic KL rows) -> decoders -> fused per-sample reconstruction sums -> ONE ELBO assembly kernel.
No host synchronisation anywhere (the reference forces >= 6 per step)."""
import os
from itertools import chain, combinations

import torch
import torch.nn.functional as F

from .. import ops
from .mmvae_base import TorchMMVAE, normal, packed_head
from .objectives import recon_rowsum


def _uses(t, stream):
    """`t` (allocated on another stream) is read by kernels queued on `stream`"""
    if t is not None and t.is_cuda and stream is not None:
        base = t._base if t._base is not None else t
        base.record_stream(stream)


class MoPOE(TorchMMVAE):
    """Generalised multimodal ELBO (mmvae_models.py:253-410).

    Parity-critical behaviour of the reference kept as is (SURVEY Appendix B + tests/golden):
      * the subset posteriors are stacked as (n_subsets, 1, B, D), so `mixture_component_selection`
        (mmvae_models.py:396-410) splits the SINGLETON axis: the joint posterior is the LAST subset -- the product
        of all modality experts and the N(0,1) prior expert -- for every sample;
      * encoder "logvar" = softmax + 1e-6 is used as log-variance inside the product and as sigma in Normal();
        the product returns the variance, used as sigma; one independent z per modality from the same joint;
      * KL terms: q_m = N(mu_m, sigma = lv_m) and the joint, against N(0, softmax(theta) * D), weights 1/(M+1).
    """

    def __init__(self, vaes, n_latents: int, obj_config: dict, model_config=None):
        super().__init__(vaes, n_latents, **obj_config)
        self.model_config = model_config
        self.modelName = "mopoe"
        self._require_normal_priors()
        self.subsets = self.set_subsets()
        self.weights = None

    @property
    def pz_params(self):
        return self._pz_params[0], F.softmax(self._pz_params[1], dim=1) * self._pz_params[1].size(-1)

    def set_subsets(self):
        """mmvae_models.py:279-294: non-empty subsets in itertools.combinations order, keyed 'mod_i_mod_j'"""
        xs = list(self.vaes.keys())
        subsets = {}
        for mod_names in chain.from_iterable(combinations(xs, n) for n in range(len(xs) + 1)):
            subsets["_".join(sorted(mod_names))] = [self.vaes[m] for m in sorted(mod_names)]
        subsets.pop("", None)
        return subsets

    # ---- hot path --------------------------------------------------------------------------------
    def _elbo_weights(self, B):
        """rows of the ELBO assembly (loss, kld) over [M recon rows, M+1 KL rows] (weighted_group_kld, objectives.py:184-201)"""
        names = list(self.vaes.keys())
        M = len(names)
        w_kl = 1.0 / (M + 1)
        lam = [float(self.vaes[n].llik_scaling) for n in names]
        return [[l / B for l in lam] + [self.obj_fn.beta * w_kl / B] * (M + 1),
                [0.0] * M + [w_kl / B] * (M + 1)]

    def _elbo_terms(self, mods, seeds=None):
        """Forward pass up to the per-row ELBO terms (mmvae_models.py:296-320 + weighted_group_kld,
        objectives.py:184-201): returns (recs, kl, W, streams, device) and leaves the tower streams un-joined.

        Stream choreography (every cross-stream wait costs ~9 us at this kernel size, a wait whose event has already
        fired costs nothing): tower i encodes on stream i; the fusion kernel runs on the LAST tower's stream -- the
        text encoder, the longer of the two -- after that stream has waited for the others; the decoders then rotate
        by one stream, so modality 0's decoder (the image ConvT stack, the longer one) continues on the fusion stream
        without a wait.  autograd replays every node on its forward stream, so the backward pass mirrors this."""
        self._begin_step()
        names = list(self.vaes.keys())
        M = len(names)
        dev = next(v["data"] for v in mods.values() if v["data"] is not None).device
        # (which tower keeps the capture stream, and the capture order of the towers, decide how hipGraph partitions the
        # step into queues: measured, see DESIGN.md section 5 -- tower 0 on the capture stream, captured first)
        main_tower = 0
        streams = self._tower_streams(dev, main_tower)
        cur = torch.cuda.current_stream(dev)
        real = [cur if st is None else st for st in streams]
        self._fork(streams, dev, mods)
        B, D = next(v["data"] for v in mods.values() if v["data"] is not None).shape[0], self.n_latents
        W = self._elbo_weights(B)
        # the M rsamples (:363-369): drawn by the fusion kernel itself unless the noise is replayed (`eps_override`)
        eps_mode = "fused"
        if self.eps_override is not None and eps_mode == "fused":
            eps_mode = "after"
        eps = self._draw_many(M, B, D, dev) if eps_mode == "first" else None
        enc = [None] * M
        enc_order = list(enumerate(zip(names, streams)))
        for i, (n, st) in enc_order:
            with torch.cuda.stream(st):
                tower = self.vaes[n].enc
                tower.raw_heads = True      # lv = softmax(u) + eta is applied by the fused latent kernel
                ops.Marks.mark(f"enc {n} start")
                try:
                    mu_lv = tower(mods[n])
                finally:
                    tower.raw_heads = False
                enc[i] = tuple(ops.mark_tensor(t, f"enc {n} out[{j}]") for j, t in enumerate(mu_lv))
                if eps is None and i == 0 and eps_mode == "after":
                    # one rsample per modality (:363-369), behind tower 0's encoder: that stream idles until the
                    # fusion anyway
                    eps = self._draw_many(M, B, D, dev)
        # the fusion runs on the last tower's stream, the decoders rotate by one stream.  (Round 6,
        # re-measured with the shorter text chain: fusion + every decoder on its OWN tower's stream -- no cross-stream wait
        # on the image chain in either direction -- replays at 0.490 ms against 0.388: hipGraph serialises that layout.)
        rotate = True
        fuse = real[-1] if rotate else real[0]
        self._fuse_stream = fuse0 = fuse
        for st in real:
            if st != fuse:
                fuse.wait_stream(st)
        theta = self._pz_params[1]
        # tensors that cross streams are registered with the consuming stream: the caching allocator must not hand
        # their memory to the producing stream again while the consumer's kernels are still queued
        for t in list(eps or []) + [t for pair in enc for t in pair]:
            _uses(t, fuse)
        with torch.cuda.stream(fuse):
            packed = [packed_head(mu, lv) for mu, lv in enc]
            if seeds is not None and getattr(self, "early_adam", None) is not None and ops.GradReducer.defer_next:
                packed[-1] = ops.EarlyStepPoint.apply(packed[-1])      # (its tower's stream IS the fusion's)
            if eps is None:
                _, kl, z = ops.poe_reparam_kl(theta, packed, M, True, (1 << (M + 1)) - 1, theta.grad, raw=True,
                                              rng=self._rng_state)
            else:
                _, kl, z = ops.poe_reparam_kl(theta, packed, eps, True, (1 << (M + 1)) - 1, theta.grad, raw=True)
        _uses(kl, cur)
        for st in real:
            if st != fuse:
                st.wait_stream(fuse)
        recs = [None] * M
        order = [(i, names[i], real[(i + M - 1) % M] if rotate else real[i]) for i in range(M)]
        for i, n, st in order:
            vae = self.vaes[n]
            _uses(z[i], st)
            with torch.cuda.stream(st):
                zi = ops.mark_tensor(z[i], f"dec {n} z")
                if seeds is not None and getattr(self, "early_adam", None) is not None and ops.GradReducer.defer_next:
                    zi = ops.DecoderEnd.apply(zi)           # (parked weight gradients of this decoder: out before the fusion)
                if seeds is not None and st != fuse0 and ops.GradReducer.side_head is not None:
                    zi = ops.StreamIdlePoint.apply(zi)      # (this decoder's stream idles behind its backward pass)
                if seeds is not None:      # the term's upstream gradient is known: its kernel also writes its backward
                    with ops.ConstSeed(seeds[i], W[0][i]):
                        # (a bce decoder is handed its target: Dec_CNN then folds the loss into its last layer's launch)
                        fuse = vae.ltype == "bce" and mods[n]["masks"] is None
                        out, _ = vae.dec({"latents": zi.unsqueeze(0), "masks": mods[n]["masks"],
                                          "bce_target": mods[n]["data"] if fuse else None})
                        r = recon_rowsum(vae.ltype, out, mods[n], laplace=self._lap(vae))
                else:
                    out, _ = vae.dec({"latents": zi.unsqueeze(0), "masks": mods[n]["masks"]})
                    r = recon_rowsum(vae.ltype, out, mods[n], laplace=self._lap(vae))
                recs[i] = ops.mark_tensor(r, f"dec {n} recon")   # (B,) = -lpx_z / llik_scaling
            _uses(recs[i], cur)
        self._fusion_inputs = packed        # the towers' packed head outputs: where the backward can be cut in two
        return recs, kl, W, streams, dev

    def objective(self, mods):
        recs, kl, W, streams, dev = self._elbo_terms(mods)
        self._join(streams, dev)
        ops.Marks.mark("joined decoders")
        out = ops.lincomb_rows(recs + [kl], W)                              # rows: M recon sums, M+1 KL rows
        return {"loss": out[0], "kld": out[1], "reconstruction_loss": recs}

    def backward_encoders(self):
        """second half of objective_backward(mods, cut=True): the encoders' backward from the saved head gradients"""
        hs, dhs, streams, dev = self._cut
        self._cut = None
        for st in streams:           # the towers' weight-gradient kernels run on their streams: the fold must join them
            if st is not None:
                ops.GradReducer.note_stream(dev, st)
        torch.autograd.backward(hs, list(dhs))
        self._join(streams, dev)

    def objective_backward(self, mods, cut=False):
        """objective(mods)["loss"].backward() without the join in the middle: the loss is LINEAR in the per-row terms
        (recs, kl) with host-side constant weights, so every tower's backward is seeded with those constants as soon as
        its own forward is done -- no tower waits for the other one's decoder or for the loss kernel.  The loss values
        are assembled afterwards, off the critical path.  Used by the captured training step (trainer.capture)."""
        first = next(v["data"] for v in mods.values() if v["data"] is not None)
        B, dev, M = first.shape[0], first.device, len(self.vaes)
        W = self._elbo_weights(B)
        key = (tuple(W[0]), B, str(dev))
        if getattr(self, "_seed_key", None) != key:      # persistent: their addresses identify them in backward
            self._seeds = [torch.full((B,), W[0][i], device=dev) for i in range(M)]
            self._seeds.append(torch.tensor(W[0][M:], device=dev).reshape(-1, 1).expand(M + 1, B).contiguous())
            self._seed_key = key
        recs, kl, W, streams, dev = self._elbo_terms(mods, self._seeds)
        def assemble(stream):
            with torch.cuda.stream(stream), torch.no_grad():
                return ops.lincomb_rows([r.detach() for r in recs] + [kl.detach()], W)

        out = None
        # the logged values ride on the end-of-backward fold launch (one extra workgroup): the forward's row sums are long
        # finished by then, and a launch of their own would sit between the fold and the optimiser (assembling them on a
        # side stream instead cost +20 us: DESIGN section 5)
        tail = ops.GradReducer.tail = ops.lincomb_rows_args(recs + [kl], W)
        early = getattr(self, "early_adam", None)
        if early is not None and not cut and ops.GradReducer.defer_next:
            # (captured one-GPU step) the decoders' + prior's parameters are updated as soon as the tower that shares the
            # fusion's stream has queued its last backward launch: ops.GradReducer.run_early_step
            ops.GradReducer.early_step = (early[0], self._fuse_stream, early[1], early[2])
            ops.GradReducer.dw_open = True
        if cut:
            # decoders + fusion only: the gradients of the towers' head outputs come back instead of flowing on
            self._cut = (list(self._fusion_inputs), torch.autograd.grad(recs + [kl], self._fusion_inputs, self._seeds),
                         streams, dev)
        else:
            torch.autograd.backward(recs + [kl], self._seeds)
        ops.GradReducer.tail = ops.GradReducer.early_step = None
        ops.GradReducer.dw_open = ops.GradReducer.early_ready = False
        if tail is not None and tail["done"]:
            out = tail["args"][2].unbind(0)
        if out is None:
            self._join(streams, dev)
            out = assemble(None)
        self._unjoined = (streams, dev)
        return {"loss": out[0], "kld": out[1], "reconstruction_loss": recs}

    def finish_step(self):
        """join the streams objective_backward() left running (call after the optimiser step has been queued)"""
        pending, self._unjoined = getattr(self, "_unjoined", None), None
        if pending is not None:
            self._join(*pending)

    # ---- API surface (inference / evaluation) ------------------------------------------------------
    def modality_mixing(self, input_batch):
        """mmvae_models.py:322-349"""
        latents = {}
        enc_mods = self.encode(input_batch)
        latents["modalities"] = enc_mods
        distr_subsets, last = {}, None
        n_all = len(self.vaes)
        for s_key, mods in self.subsets.items():
            if all(m.modelName in input_batch and input_batch[m.modelName]["data"] is not None for m in mods):
                mus = [enc_mods[m.modelName]["shared"][0] for m in mods]
                lvs = [enc_mods[m.modelName]["shared"][1] for m in mods]
                s_mu, s_var = self.product_of_experts(mus, lvs, with_prior=(len(mods) == n_all))   # :385-389
                distr_subsets[s_key] = [s_mu.unsqueeze(0), s_var.unsqueeze(0)]
                last = (s_mu, s_var)
        self.weights = (1 / float(len(distr_subsets))) * torch.ones(len(distr_subsets), device=last[0].device)
        latents["joint"] = [last[0], last[1]]        # singleton-axis selection => last available subset (:396-410)
        latents["subsets"] = distr_subsets
        return latents

    def forward(self, inputs, K=1):
        """mmvae_models.py:351-370"""
        latents = self.modality_mixing(inputs)
        qz_d, px_d, z_d, qz_joint = {}, {}, {}, {}
        j_mu, j_var = latents["joint"]
        for mod, vae in self.vaes.items():
            sh = latents["modalities"][mod]["shared"]
            qz_d[mod] = normal(*sh) if sh is not None else None
            qz_joint[mod] = normal(j_mu, j_var)
            eps = torch.stack([self._draw(j_mu.shape[0], j_mu.shape[1], j_mu.device) for _ in range(K)])
            z = j_mu + j_var * eps
            z_d[mod] = {"latents": z, "masks": inputs[mod]["masks"]}
            px_d[mod] = self._px(vae, *vae.dec(z_d[mod]))
        return self.make_output_dict(qz_d, px_d, z_d, qz_joint)


class POE(TorchMMVAE):
    """MVAE, product of experts (mmvae_models.py:134-250)."""

    # decoder calls of one step spread over the two streams by their launch counts (_decoder_lanes); MMVAE_POE_BALANCE=0:
    # every call on its tower's stream
    balance_decoder_calls = os.environ.get("MMVAE_POE_BALANCE", "1") != "0"

    def __init__(self, vaes, n_latents: int, obj_config: dict, model_config=None):
        super().__init__(vaes, n_latents, **obj_config)
        self.model_config = model_config
        self.modelName = "poe"
        for vae in self.vaes.values():
            assert vae.prior_str in ["normal", "gaussian"], "POE only works with gaussian priors! Adjust the config"
        # utils.subsample_input_modalities (utils.py:86-112) iterates a Python set => its order within one subset
        # size depends on PYTHONHASHSEED; the default here is itertools order, `subset_order` pins another one.
        self.subset_order = None
        self.batch_dropout_towers = True      # decoders with dropout: all subset passes in one call (objective docstring)
        self._job_calls = {}                  # launches of a decoder call the last time it ran (_decoder_lanes)

    @property
    def pz_params(self):
        return self._pz_params[0], F.softmax(self._pz_params[1], dim=1) * self._pz_params[1].size(-1)

    def _subsets(self):
        names = list(self.vaes.keys())
        if self.subset_order is not None:
            return [tuple(names[i] for i in s) for s in self.subset_order]
        return [c for n in range(1, len(names) + 1) for c in combinations(names, n)]

    def objective(self, mods):
        """mmvae_models.py:159-187: sum over input subsets of -(sum_b sum_m lpx - beta sum_b KL(joint || prior)).

        The reference runs the whole model once per subset (2^M - 1 passes: every encoder 2^(M-1) times on the SAME
        input, every decoder 2^M - 1 times).  Round 4: a tower without dropout (the CNN image towers; any tower in eval
        mode) takes all of its passes in ONE call -- the encoder's output is simply shared by the subsets it belongs to,
        the decoder decodes the (n_subsets * B) latent samples as one batch and the row-sum kernels pair output row r with
        target row r % B.  Same sums, same noise draws in the same order.  Decoders WITH dropout do the same
        (`batch_dropout_towers`: their masks are then drawn once for all passes); encoders with dropout keep one call per
        subset -- the text encoder's positional encoding is indexed by the BATCH position (nn_modules.py:430-438), which a
        repeated batch would change.  BASELINE configs[0]: 181 -> 96 C-ABI calls per step."""
        self._begin_step()
        names = list(self.vaes.keys())
        M = len(names)
        theta = self._pz_params[1]
        subsets = self._subsets()
        NS = len(subsets)
        dev = next(v["data"] for v in mods.values() if v["data"] is not None).device
        streams = self._tower_streams(dev)

        def one_call(part):      # every pass of this tower gives the same function of its input
            return getattr(part, "drop_state", None) is None or not part.training

        # ---- encoders: packed[s][n] for n in subset s ----
        packed = [dict() for _ in subsets]
        raw_heads = dev.type == "cuda" and all(hasattr(v.enc, "raw_heads") and v.enc.enc_mu_logvar for v in self.vaes.values())
        self._fork(streams, dev, mods)
        for n, st in zip(names, streams):
            member = [s for s, S in enumerate(subsets) if n in S]
            with torch.cuda.stream(st):
                enc = self.vaes[n].enc
                # lv = softmax(u) + eta is applied by the fusion kernels (as MoPOE._elbo_terms does): no softmax launch
                # behind the heads, none in front of their backward
                enc.raw_heads = raw_heads
                try:
                    if one_call(enc):
                        p = packed_head(*enc(mods[n]))
                        for s in member:
                            packed[s][n] = p
                    elif self.batch_dropout_towers and len(member) > 1 and getattr(enc, "takes_repeat", False) and \
                            enc.repeat_ok(mods[n]):
                        # an encoder WITH dropout whose passes differ only in their masks: all of them as one call over
                        # len(member) * B sequences (round 5: the text encoder's positional term follows the sample's
                        # position in the ORIGINAL batch, ops.embed_pe), rows k * B + b
                        p = packed_head(*enc(dict(mods[n], repeat=len(member))))
                        for s, part in zip(member, ops.split_rows(p, len(member))):      # (row blocks; ONE cat in backward)
                            packed[s][n] = part
                    else:
                        for s in member:
                            packed[s][n] = packed_head(*enc(mods[n]))
                finally:
                    enc.raw_heads = False
        self._join(streams, dev)
        # Tensors that cross streams (a tower's heads -> the fusion; the latent samples -> decoder calls on the other stream;
        # their row sums -> the ELBO assembly) are registered with the consuming stream.  Without it the allocator hands a
        # block out again as soon as the host drops the last reference -- found in round 6 in the CAPTURED step: the
        # image-only subset's z (capture stream) was re-used by the image decoder's backward while the text decoder's
        # cross-attention weight gradient, queued long before on the side stream, had yet to read it.
        cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        for d_ in packed:
            for t in d_.values():
                _uses(t, cur)
        # ---- fusion per subset (one noise draw each, in subset order) ----
        zs, kl_blocks = [], []
        D = self.n_latents
        B = next(iter(packed[0].values())).shape[0]
        draws = self._draw_many(NS, B, D, dev)      # (one launch; `eps_override`: NS recorded draws in subset order)
        for s, S in enumerate(subsets):
            ps = [packed[s][n] for n in names if n in S]
            eps = [draws[s]]
            E = len(ps)
            _, kl, z = ops.poe_reparam_kl(theta, ps, eps, True, 1 << E, theta.grad, raw=raw_heads)
            zs.append(z[0])
            kl_blocks.append(kl)             # (E + 1, B): the joint's KL is the last row
        # ---- decoders: one (passes, B) block of row sums per call ----
        # One call for all passes of a decoder that run under the SAME mask: the B-row latent samples of those subsets are
        # one batch (row k * B + b).  Subsets that contain the modality decode under its mask, the others under `masks=None`
        # (the decoder's full-length all-ones mask, another sequence length for the text tower): two groups.
        # optimal_sigma fits ONE sigma per call, so its passes cannot share one.  A decoder WITH dropout then draws its
        # masks once per group instead of per pass (independent masks either way; the extracted-mask parity test splits
        # them by pass).
        jobs = []      # (modality index, name, subsets of the call, mask of the call, (slot, call) of its dropout state)
        for i, n in enumerate(names):
            vae = self.vaes[n]
            mk = mods[n]["masks"]
            share = vae.ltype != "optimal_sigma" and (one_call(vae.dec) or self.batch_dropout_towers)
            groups = [[s_ for s_, S in enumerate(subsets) if n not in S], [s_ for s_, S in enumerate(subsets) if n in S]]
            if mk is None:
                groups = [list(range(NS))]
            groups = sorted([g for g in groups if g], key=lambda g: g[0])          # call order = first member's order
            for g in groups:
                gm = mk if (mk is not None and n in subsets[g[0]]) else None
                for part in ([g] if share and len(g) > 1 else [[s_] for s_ in g]):
                    jobs.append((i, n, part, gm))
        lanes = self._decoder_lanes(jobs, streams, B)
        # every call's input batch -- the samples of its subsets as row blocks -- from ONE launch, and in backward every
        # sample's gradient as one sum over the calls that decoded it (ops.RowsFan), instead of a torch.cat per call and an
        # addition per sample
        plan = [(len(g), D, [(s_, k, 0) for k, s_ in enumerate(g)]) for _, _, g, _ in jobs]
        z_ins = ops.rows_fan(plan, zs) if (zs[0].is_cuda and ops.RowsFan.supported(plan, zs)) else None
        rec_blocks = []
        self._fork(streams, dev, mods)
        for j_, ((i, n, g, gm), (st, begun)) in enumerate(zip(jobs, lanes)):
            vae = self.vaes[n]
            with torch.cuda.stream(st):
                c0 = ops.CALLS[0]
                m_in = gm if (gm is None or len(g) == 1) else gm.repeat(len(g), 1)
                if z_ins is not None:
                    z_in = z_ins[j_]
                    _uses(z_in, st)
                else:
                    for s_ in g:
                        _uses(zs[s_], st)
                    z_in = torch.cat([zs[s_] for s_ in g], 0) if len(g) > 1 else zs[g[0]]
                job = {"latents": z_in.unsqueeze(0), "masks": m_in}
                if begun is not None:
                    job["drop_begun"] = begun
                if gm is None and mods[n]["masks"] is not None and getattr(vae.dec, "takes_keep_steps", False):
                    job["keep_steps"] = int(mods[n]["masks"].shape[1])      # (recon_rowsum slices to the target's mask length)
                out, _ = vae.dec(job)
                r = recon_rowsum(vae.ltype, out, mods[n])                    # target row = output row % B
                _uses(r, cur)
                rec_blocks.append((r.view(len(g), B), i, list(g)))           # the call's rows as ONE (passes, B) block
                self._job_calls[(n, len(g), gm is None, B)] = ops.CALLS[0] - c0
        self._join(streams, dev)
        # ELBO assembly straight on the calls' row BLOCKS (round 5: slicing the batched row sums and selecting the joint KL
        # row cost ~40 ATen fill / copy / add launches per step in forward + autograd): every block is addressed in place,
        # a row that does not enter an output has weight 0.  Outputs: loss, kld and -- while they fit (<= 4 outputs) -- the
        # logged per-modality reconstruction sums of each modality's own singleton subset (mmvae_models.py:181-182).
        blocks, W_loss, W_kld = [], [], []
        W_ind = [[] for _ in range(M)]
        n_rows = sum(b.shape[0] for b, _, _ in rec_blocks) + sum(k.shape[0] for k in kl_blocks)
        whole_kl = n_rows <= 32
        for blk, i, subs in rec_blocks:
            blocks.append(blk)
            for s_ in subs:
                W_loss.append(float(self.vaes[names[i]].llik_scaling))
                W_kld.append(0.0)
                for j in range(M):
                    W_ind[j].append(1.0 if (j == i and s_ == i) else 0.0)
        for s_, kl in enumerate(kl_blocks):
            E1 = kl.shape[0]
            if whole_kl:
                blocks.append(kl)
                W_loss += [0.0] * (E1 - 1) + [float(self.obj_fn.beta)]
                W_kld += [0.0] * (E1 - 1) + [1.0 / NS]
                for j in range(M):
                    W_ind[j] += [0.0] * E1
            else:
                blocks.append(kl[E1 - 1])
                W_loss.append(float(self.obj_fn.beta))
                W_kld.append(1.0 / NS)
                for j in range(M):
                    W_ind[j].append(0.0)
        if M <= 2 and len(W_loss) <= 32:
            out = ops.lincomb_rows(blocks, [W_loss, W_kld] + W_ind)
            ind = [o.detach() for o in out[2:]]
        else:
            out = ops.lincomb_rows(blocks, [W_loss, W_kld])
            ind = [None] * M
            for blk, i, subs in rec_blocks:
                for k, s_ in enumerate(subs):
                    if s_ == i:
                        ind[i] = blk[k].detach().sum()
        return {"loss": out[0], "reconstruction_loss": ind, "kld": out[1]}

    def _decoder_lanes(self, jobs, streams, B):
        """Stream of every decoder call of a step, [(stream, (slot, call) | None)] in call order.

        By default a call runs on its tower's stream.  Two streams, several calls: the calls are spread over the two by the
        number of launches each one took the last time it ran (longest first onto the shorter lane) -- BASELINE configs[0]
        decodes the text once at full length for the image-only subset (45 steps: the layer's launch-per-op form, ~20
        launches each way) and once under the mask for the other two (one launch per layer); behind one another on the text
        tower's stream they are the step's critical chain while the image decoder's stream idles.  A decoder with dropout
        whose calls leave its tower's stream gets its per-call counter advances up front, in call order, on the caller's
        stream (DropoutState.begin launches on the current stream: two lanes would race on the counter)."""
        own = [(streams[i], None) for i, _, _, _ in jobs]
        self.decoder_calls_moved = 0      # (tests: calls of the last step that left their tower's stream)
        lanes = sorted(set(streams), key=lambda s: s is not None)
        if not POE.balance_decoder_calls or len(lanes) != 2 or len(jobs) < 3:
            return own
        cost = [self._job_calls.get((n, len(g), gm is None, B)) for i, n, g, gm in jobs]
        if any(c is None for c in cost):
            return own
        load = {lanes[0]: 0, lanes[1]: 0}
        pick = [None] * len(jobs)
        for k in sorted(range(len(jobs)), key=lambda k: -cost[k]):
            st = min(lanes, key=lambda s: (load[s], s is not None))
            pick[k] = st
            load[st] += cost[k]
        out = []
        for k, (i, n, g, gm) in enumerate(jobs):
            dec = self.vaes[n].dec
            ds = getattr(dec, "drop_state", None)
            moved = any(pick[j] is not streams[i] for j, job in enumerate(jobs) if job[0] == i)
            begun = ds.begin() if (ds is not None and dec.training and moved) else None
            out.append((pick[k], begun))
            self.decoder_calls_moved += pick[k] is not streams[i]
        return out

    def modality_mixing(self, x):
        """mmvae_models.py:210-232: prior expert + present encoders -> product"""
        mus, lvs, single = [], [], {}
        for m, vae in self.vaes.items():
            if x[m]["data"] is not None:
                mu, lv = vae.enc(x[m])
                single[m] = normal(mu, lv)
                mus.append(mu)
                lvs.append(lv)
        mu, var = self.product_of_experts(mus, lvs, with_prior=True)
        return mu, var, single

    def forward(self, inputs, K=1):
        """mmvae_models.py:189-208"""
        mu, var, single = self.modality_mixing(inputs)
        qz_x = normal(mu, var)
        eps = torch.stack([self._draw(mu.shape[0], mu.shape[1], mu.device) for _ in range(K)])
        z = mu + var * eps
        qz_d, px_d, z_d = {}, {}, {}
        for mod, vae in self.vaes.items():
            px_d[mod] = normal(*vae.dec({"latents": z, "masks": inputs[mod]["masks"]}))
        for key in inputs.keys():
            qz_d[key] = qz_x
            z_d[key] = {"latents": z, "masks": inputs[key]["masks"]}
        return self.make_output_dict(single, px_d, z_d, joint_dist=qz_d)


class MOE(TorchMMVAE):
    """MMVAE, mixture of experts (mmvae_models.py:10-131): objective "elbo" with K = 1, "dreg"
    with any K on towers that keep the K axis (the shipped configs/config_mnistsvhn.yml: MNIST / SVHN towers, K = 30,
    `prior: laplace`) and "iwae" with any K on any towers (literal where the reference runs -- B = 1 or K = 1 --, a
    defined extension with the K-preserving text decoder beyond: MultimodalObjective.iwae; BASELINE configs[2] =
    iwae, K = 8, CdSprites+ towers).  elbo with K > 1 fails in the reference.

    q_m = Normal | Laplace(mu_m, scale = lv_m) -- the config's `prior` key also names the posterior and the likelihood
    family (models/trainer.py:104) --, K samples z_m per modality; every modality is decoded from its own z
    (likelihood Normal, :101-103) and from the z of the LAST other modality (likelihood `vae.px_z`, the dict overwrite
    of :112-116).
    elbo: KL against the per-VAE fixed N(0,1) prior through the model-level `self.pz` = Normal (:45); cross terms
      weighted by exp(log q_r(z_o) - log q_o(z_o).detach()) with z_o detached; rows whose weighted sum is exactly 0
      are dropped and beta*kld.sum() is subtracted once per surviving row (:73, objectives.py:67); loss / M.
    dreg: objectives.py:361-387 with the TRAINABLE model prior N(0, softmax(theta) D) (:76, pz_params)."""

    def __init__(self, vaes, n_latents: int, obj_config: dict, model_config=None):
        super().__init__(vaes, n_latents, **obj_config)
        self.model_config = model_config
        self.modelName = "moe"
        obj = self.obj_fn.obj_name
        if obj == "elbo" and self.K != 1:
            raise NotImplementedError("moe: obj elbo with K > 1 fails in the reference (mmvae_models.py:62); use dreg")
        if obj in ("dreg", "iwae") and len(self.vaes) != 2:
            raise NotImplementedError("moe dreg / iwae: the reference's cross-term indexing (mmvae_models.py:64-70) is "
                                      "only meaningful for two modalities")
        for vae in self.vaes.values():
            if vae.prior_str not in ("normal", "gaussian", "laplace"):
                raise NotImplementedError(f"prior: {vae.prior_str} is not on the MI355X path (normal, laplace are)")
        self._laplace = [v.prior_str == "laplace" for v in self.vaes.values()]
        self.register_buffer("_theta0", torch.zeros(1, n_latents), persistent=False)   # softmax(0)*D = 1: N(0,1)

    @property
    def pz_params(self):
        return self._pz_params[0], F.softmax(self._pz_params[1], dim=1) * self._pz_params[1].size(-1)

    batch_passes = True      # elbo: a decoder's own + cross pass in one call (objective)

    def _draw_k(self, m, K, B, D, dev):
        """(K,B,D) standard variates of q_m's family: one recorded draw (`eps_override`) or the device generator"""
        if self.eps_override is not None:
            return self.eps_override.pop(0).reshape(K, B, D).to(device=dev, dtype=torch.float32).contiguous()
        return (ops.rand_laplace if self._laplace[m] else ops.randn)((K, B, D), self._rng_state)

    def objective(self, data):
        """mmvae_models.py:32-78"""
        if self.obj_fn.obj_name in ("dreg", "iwae"):        # same forward; the objective differs (objectives.py)
            return self._objective_dreg(data)
        self._begin_step()
        names = list(self.vaes.keys())
        M = len(names)
        dev = next(v["data"] for v in data.values() if v["data"] is not None).device
        # Round 4: the towers on two streams (encoders side by side, decoders + their loss terms side by side; autograd
        # replays every node on its forward stream), and each decoder's own + cross pass as ONE call over 2 B latent
        # samples (rows [0, B) decode z_r, rows [B, 2B) z_o; the row-sum kernels pair output row k with target row k % B).
        # Same sums; a decoder with dropout draws its masks once for both passes (independent masks either way: the
        # extracted-mask parity test splits them by pass).  optimal_sigma fits ONE sigma per call: its passes stay apart.
        # The shipped config_cdspritesplus.yml step is a chain of ~260 launches, ~115 of them the text decoder's two
        # op-by-op passes: 4.00 -> 3.4 ms/step.
        streams = self._tower_streams(dev)
        cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        real = [cur if st is None else st for st in streams]
        self._fork(streams, dev, data)
        packed = [None] * M
        for i, (n, st) in enumerate(zip(names, streams)):
            with torch.cuda.stream(st):
                packed[i] = packed_head(*self.vaes[n].enc(data[n]))
        self._join(streams, dev)
        B, D = packed[0].shape[0], self.n_latents
        zs, kls = [], []
        for i in range(M):
            eps = self._draw_k(i, 1, B, D, dev).reshape(B, D)
            if self._laplace[i]:      # z = mu + scale * e all the same; KL(Laplace || N(0,1)) has its own closed form
                _, _, z = ops.poe_reparam_kl(self._theta0, [packed[i]], [eps], 2, 0)
                kls.append(ops.kl_laplace_normal(packed[i]))
            else:
                _, kl, z = ops.poe_reparam_kl(self._theta0, [packed[i]], [eps], 2, 0b10)
                kls.append(ops.kl_row(kl, 1))      # (kl_mask 0b10: row 1 is the only one the backward reads)
            zs.append(z[0])
        for t in packed + zs:         # read by both towers' decoder sides
            for st in real:
                _uses(t, st)
        rows, W = [None] * (2 * M), []
        self._fork(streams, dev, data)
        for r, (n, st) in enumerate(zip(names, streams)):
            vae = self.vaes[n]
            o = [s for s in range(M) if s != r][-1]
            with torch.cuda.stream(st):
                mk = data[n]["masks"]
                if vae.ltype != "optimal_sigma" and self.batch_passes:
                    z2 = torch.cat([zs[r], zs[o]], 0)
                    out, _ = vae.dec({"latents": z2.unsqueeze(0), "masks": None if mk is None else mk.repeat(2, 1)})
                    # own: dist.Normal(*px_z) (:101-103); cross (rows B..2B): vae.px_z = the config's `prior` family (:115)
                    rs = recon_rowsum(vae.ltype, out, data[n], laplace=(0b10, B) if self._laplace[r] else False)
                    own_r, cross_r = rs.view(2, B).unbind(0)      # (backward: one stack instead of two zero-fill + copy pairs)
                else:
                    own, _ = vae.dec({"latents": zs[r].unsqueeze(0), "masks": mk})
                    cross, _ = vae.dec({"latents": zs[o].unsqueeze(0), "masks": mk})
                    own_r = recon_rowsum(vae.ltype, own, data[n])
                    cross_r = recon_rowsum(vae.ltype, cross, data[n], laplace=self._laplace[r])
                ratio = ops.laplace_logratio if self._laplace[r] else ops.normal_logratio
                lw = ratio(packed[r], packed[o].detach(), zs[o].detach())
                rows[2 * r], rows[2 * r + 1] = own_r, ops.expmul(lw, cross_r)
            for t in (rows[2 * r], rows[2 * r + 1]):
                _uses(t, cur)
            W += [float(vae.llik_scaling)] * 2
        self._join(streams, dev)
        kld = torch.stack(kls)                                              # (M, B), also the logged "kld"
        loss = ops.moe_elbo(rows, W, kld, self.obj_fn.beta, M)
        with torch.no_grad():                                               # logged only: lpx rows, reference sign
            lpx = list(torch._foreach_mul([r.detach() for r in rows], [-w for w in W]))      # (one launch)
        return {"loss": loss, "reconstruction_loss": lpx, "kld": kld}

    def _objective_dreg(self, data):
        """MOE.objective's non-elbo branch + MultimodalObjective.dreg / .iwae (mmvae_models.py:63-78,
        objectives.py:342-387).  Every decoder decodes ALL M*K*B latent samples in one pass (rows [r*K*B, (r+1)*K*B) are
        z_r): its own block is the own reconstruction, the other block the cross reconstruction.  dreg: `lprob` towers
        that keep K (its lw sums over the batch per k, which only those towers' (K,) sums allow); iwae: any tower, the
        target rows repeat over (r, k) inside the loss kernels (row % B)."""
        self._begin_step()
        names = list(self.vaes.keys())
        M, K, D = len(names), int(self.K), self.n_latents
        dev = next(v["data"] for v in data.values() if v["data"] is not None).device
        streams = self._tower_streams(dev)
        cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        real = [cur if st is None else st for st in streams]
        self._fork(streams, dev, data)
        packed = [None] * M
        for i, (n, st) in enumerate(zip(names, streams)):
            with torch.cuda.stream(st):
                packed[i] = packed_head(*self.vaes[n].enc(data[n]))
        self._join(streams, dev)
        B = packed[0].shape[0]
        eps = [self._draw_k(m, K, B, D, dev) for m in range(M)]
        theta = self._pz_params[1]
        iwae = self.obj_fn.obj_name == "iwae"
        # lat = log p(z) - beta lqz: dreg has no beta (objectives.py:372), iwae multiplies lqz by it (:356)
        lat, z = ops.moe_ksample(theta, packed, eps, self._laplace, theta.grad,
                                 beta=float(self.obj_fn.beta) if iwae else 1.0)          # z: (M,K,B,D)
        KB = K * B
        rows, lam = [None] * (2 * M), []
        for st in real:
            _uses(z, st)
        self._fork(streams, dev, data)
        for r, (n, st) in enumerate(zip(names, streams)):
            vae = self.vaes[n]
            if not iwae and (vae.ltype != "lprob" or data[n]["masks"] is not None):
                raise NotImplementedError("moe dreg: recon_loss lprob on unmasked modalities (the MNIST / SVHN towers) "
                                          "is what keeps the K axis in the reference")
            self.obj_fn.set_ltype(vae.ltype)
            o = 1 - r
            with torch.cuda.stream(st):
                out, _ = vae.dec({"latents": z.view(M * K, B, D), "masks": data[n]["masks"]})     # (M*K, B, ...) / (M*K*B, ...)
                # own block r: dist.Normal (:101-103); cross block o: vae.px_z = the config's `prior` family (:115)
                lap_mask = (1 << o) if self._laplace[r] else 0
                if vae.ltype == "optimal_sigma":
                    # optimal_sigma fits ONE sigma per reference call, and the reference evaluates the own and the cross
                    # reconstruction in separate calls (mmvae_models.py:63-78): row sums per source posterior (found by
                    # the K-sample test on the action towers, round 5: one sigma over all M K B rows was 9e-4 off)
                    flat = out.reshape(M, KB, *out.shape[-3:]) if out.dim() >= 4 else out.reshape(M, KB, -1)
                    blocks = [recon_rowsum(vae.ltype, flat[m], data[n]) for m in range(M)]
                    rs = blocks[0]
                    for blk in blocks:
                        _uses(blk, cur)
                else:
                    rs = recon_rowsum(vae.ltype, out, data[n], laplace=(lap_mask, KB))          # (M*K*B,)
                    blocks = rs.view(M, KB).unbind(0)
                rows[2 * r], rows[2 * r + 1] = blocks[r], blocks[o]
            _uses(rs, cur)
            lam.append(float(vae.llik_scaling))
        self._join(streams, dev)
        return self.obj_fn.calculate_loss({"lat": lat, "rows": rows, "lam": lam})

    def modality_mixing(self, mods):
        return self.encode(mods)

    def forward(self, x, K=1):
        """mmvae_models.py:80-117, including the cross-generation calls with missing modalities
        (`data` None, masks kept: models/trainer.py:179-215): a missing modality takes the latent sample of the FIRST
        present one (the reference aliases that modality's dict, :105-108) and is decoded from it under its own masks;
        every target's cross entry is a fresh one-entry dict, so the last source in dict order wins (:109-114)."""
        missing, filled = self.get_missing_modalities(x)
        assert len(filled) > 0, "at least one modality must be present for forward call"
        qz, zs, px, cross = {}, {}, {}, {}
        for m, vae in self.vaes.items():
            if x[m]["data"] is None:
                qz[m] = None
                continue
            mu, lv = vae.enc(x[m])
            qz[m] = normal(mu, lv)
            eps = torch.stack([self._draw(mu.shape[0], mu.shape[1], mu.device) for _ in range(K)])
            zs[m] = {"latents": mu + lv * eps, "masks": x[m]["masks"]}
        for m in missing:
            zs[m] = {"latents": zs[filled[0]]["latents"], "masks": x[m]["masks"]}
        zs = {m: zs[m] for m in self.vaes}                     # modality order, as the reference's dict
        for m, vae in self.vaes.items():
            px[m] = normal(*vae.dec({"latents": zs[m]["latents"], "masks": x[m]["masks"]}))
        for src, z in zs.items():
            for tgt, vae in self.vaes.items():
                if tgt != src:
                    cross[tgt] = {src: normal(*vae.dec({"latents": z["latents"], "masks": x[tgt]["masks"]}))}
        return self.make_output_dict(qz, px, zs, cross_decoder_dist=cross)


class DMVAE(TorchMMVAE):
    """DMVAE, shared + private latents (mmvae_models.py:413-509), K = 1, all modalities present.

    Every encoder emits D + P columns; [:D] is the shared posterior, [D:] the private one (split after the softmax
    over all D + P columns, mmvae_base.py:152-156).  joint = product of the shared experts, no prior expert.  Noise is
    drawn in the reference's order: z_joint, then per modality z_shared, z_private and a fresh shared draw of every
    other modality for the cross reconstruction.  Three ELBOs per modality (:458-459), all summed over the batch."""

    def __init__(self, vaes, n_latents: int, obj_config: dict, model_config=None):
        super().__init__(vaes, n_latents, **obj_config)
        self.model_config = model_config
        self.modelName = "dmvae"
        self._require_normal_priors()
        assert self.latent_factorization, "DMVAE requires private_latents in the config"
        if self.K != 1:
            raise NotImplementedError("dmvae: K = 1 only on this path")
        pmax = max(v.private_latents for v in self.vaes.values())
        self.register_buffer("_theta0", torch.zeros(1, pmax), persistent=False)    # N(0,1) prior of the private part

    @property
    def pz_params(self):
        return self._pz_params[0], F.softmax(self._pz_params[1], dim=1) * self._pz_params[1].size(-1)

    batch_passes = True      # a decoder's own + joint + cross passes in one call (objective)

    def objective(self, mods):
        self._begin_step()
        names = list(self.vaes.keys())
        M, D = len(names), self.n_latents
        beta = float(self.obj_fn.beta)
        theta = self._pz_params[1]
        dev = next(v["data"] for v in mods.values() if v["data"] is not None).device
        # Round 4: towers on two streams (as MOE.objective), and a decoder's passes -- own, joint, cross (one per other
        # modality) -- as ONE call over (2 + (M - 1)) B latent samples (the row-sum kernels pair output row k with target
        # row k % B; a decoder with dropout draws its masks once for all passes; optimal_sigma fits one sigma per call:
        # its passes stay apart).  BASELINE configs[3]: 1.54 -> 1.2 ms/step.
        streams = self._tower_streams(dev)
        cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        real = [cur if st is None else st for st in streams]
        self._fork(streams, dev, mods)
        packed = [None] * M
        for i, (n, st) in enumerate(zip(names, streams)):
            with torch.cuda.stream(st):
                packed[i] = packed_head(*self.vaes[n].enc(mods[n]))
        self._join(streams, dev)
        B = packed[0].shape[0]
        P = [self.vaes[n].private_latents for n in names]
        # noise in the reference's draw order (mmvae_models.py:486-502)
        e_sh, e_pr, e_cross = {}, {}, {}
        if self.eps_override is None and len(set(P)) == 1:
            # two launches instead of 1 + M (M + 1): every D-wide draw of the step from one (n, B, D) block, the private
            # ones from one (M, B, P) block (independent standard normals either way; recorded noise keeps the order below)
            wide = self._draw_many(1 + M * M, B, D, dev)
            priv = self._draw_many(M, B, P[0], dev)
            e_joint, k = wide[0], 1
            for i in range(M):
                e_sh[i], e_pr[i], k = wide[k], priv[i], k + 1
                for m in range(M):
                    if m != i:
                        e_cross[(i, m)], k = wide[k], k + 1
        else:
            e_joint = self._draw(B, D, dev)
            for i in range(M):
                e_sh[i] = self._draw(B, D, dev)
                e_pr[i] = self._draw(B, P[i], dev)
                for m in range(M):
                    if m != i:
                        e_cross[(i, m)] = self._draw(B, D, dev)        # target i, fresh draw of q_shared(m)
        _, klj, zj = ops.poe_reparam_kl(theta, packed, [e_joint], 0, 1 << M, theta.grad, cols=(0, D))
        z_sh, z_cr, kl_sh, z_pr, kl_pr = {}, {}, {}, {}, {}
        for m in range(M):
            targets = [i for i in range(M) if i != m]
            _, kl, z = ops.poe_reparam_kl(theta, [packed[m]], [e_sh[m]] + [e_cross[(i, m)] for i in targets], 2, 0b10,
                                          theta.grad, cols=(0, D))
            z_sh[m], kl_sh[m] = z[0], kl                    # (kl: (2, B), the posterior's KL is row 1)
            for i, zc in zip(targets, z[1:]):
                z_cr[(i, m)] = zc
            _, klp, zp = ops.poe_reparam_kl(self._theta0[:, :P[m]].contiguous(), [packed[m]], [e_pr[m]], 2, 0b10, None,
                                            cols=(D, P[m]))
            z_pr[m], kl_pr[m] = zp[0], klp
        for t in [zj[0]] + list(z_sh.values()) + list(z_cr.values()) + list(z_pr.values()):
            for st in real:
                _uses(t, st)
        per = [None] * M
        # every decoder's input batch -- rows [own | joint | cross...] x columns [shared | private] -- from ONE launch for all
        # modalities, and every sample's gradient as one sum in backward (ops.RowsFan): 2 cat + 1 repeat per modality forward,
        # a repeat-sum, the strided slices' copies and an addition per shared sample backward before
        batched = [self.vaes[n].ltype != "optimal_sigma" and self.batch_passes for n in names]
        lats = None
        if all(batched) and zj[0].is_cuda:
            srcs, index, plan = [], {}, []

            def src(t):
                if id(t) not in index:
                    index[id(t)] = len(srcs)
                    srcs.append(t)
                return index[id(t)]
            for i in range(M):
                zs_i = [z_sh[i], zj[0]] + [z_cr[(i, m)] for m in range(M) if m != i]
                plan.append((len(zs_i), D + P[i], [(src(z), k, 0) for k, z in enumerate(zs_i)] +
                             [(src(z_pr[i]), k, D) for k in range(len(zs_i))]))
            if ops.RowsFan.supported(plan, srcs):
                lats = ops.rows_fan(plan, srcs)
        self._fork(streams, dev, mods)
        for i, (n, st) in enumerate(zip(names, streams)):
            vae = self.vaes[n]
            with torch.cuda.stream(st):
                zs_i = [z_sh[i], zj[0]] + [z_cr[(i, m)] for m in range(M) if m != i]      # decode order own, joint, cross
                mk = mods[n]["masks"]                                                  # (mmvae_models.py:494-502)
                if vae.ltype != "optimal_sigma" and self.batch_passes:
                    NP = len(zs_i)
                    if lats is not None:
                        lat = lats[i]
                        _uses(lat, st)
                    else:
                        lat = torch.cat([torch.cat(zs_i, 0), z_pr[i].repeat(NP, 1)], -1)
                    out, _ = vae.dec({"latents": lat.unsqueeze(0), "masks": None if mk is None else mk.repeat(NP, 1)})
                    rs = recon_rowsum(vae.ltype, out, mods[n], laplace=self._lap(vae))
                    per[i] = [rs.view(NP, B)]                  # ONE (passes, B) block: rows own, joint, cross...
                else:
                    per[i] = []
                    for z in zs_i:
                        out, _ = vae.dec({"latents": torch.cat([z, z_pr[i]], -1).unsqueeze(0), "masks": mk})
                        per[i].append(recon_rowsum(vae.ltype, out, mods[n], laplace=self._lap(vae)).view(1, B))
            for t in per[i]:
                _uses(t, cur)
        self._join(streams, dev)
        # ELBO assembly on the calls' row BLOCKS (as POE.objective: selecting rows out of the batched row sums / the KL
        # blocks cost a zero-fill and a copy per row in autograd's select / unbind backward -- 15 launches per step at
        # M = 2): every block is addressed in place, a row that does not enter an output has weight 0
        blocks, W_loss, W_kld = [], [], []
        W_ind = [[] for _ in range(M)]

        def put(blk, wl, wk, ind_of=None):
            blocks.append(blk)
            W_loss.extend(wl)
            W_kld.extend(wk)
            for j in range(M):
                W_ind[j].extend([1.0 if (j == ind_of and r == 0) else 0.0 for r in range(len(wl))])
        for i, n in enumerate(names):
            lam = float(self.vaes[n].llik_scaling)
            first = True
            for blk in per[i]:                                  # rows in decode order: own, joint, cross...
                r = blk.shape[0]
                put(blk, [lam] * r, [0.0] * r, i if first else None)
                first = False
            put(kl_sh[i], [0.0, beta], [0.0, 1.0 / M])
            put(kl_pr[i], [0.0, beta * (M - 1)], [0.0, 0.0])
        put(klj, [0.0] * M + [beta * M], [0.0] * (M + 1))       # the joint's KL (last row) enters every modality's joint ELBO
        if len(W_loss) <= 32 and M <= 2:
            out = ops.lincomb_rows(blocks, [W_loss, W_kld] + W_ind)
            ind = [o.detach() for o in out[2:]]
        elif len(W_loss) <= 32:
            out = ops.lincomb_rows(blocks, [W_loss, W_kld])
            ind = [per[i][0][0].detach().sum() for i in range(M)]
        else:       # (more rows than one assembly launch addresses: the selected rows only)
            rows, W_loss, W_kld = [], [], []
            for i, n in enumerate(names):
                lam = float(self.vaes[n].llik_scaling)
                flat_rows = [r_ for blk in per[i] for r_ in blk.unbind(0)]
                own, joint, cross = flat_rows[0], flat_rows[1], flat_rows[2:]
                rows += [own, kl_sh[i][1], joint, klj[M]] + cross + [kl_pr[i][1]]
                W_loss += [lam, beta, lam, beta] + [lam] * len(cross) + [beta * len(cross)]
                W_kld += [0.0, 1.0 / M, 0.0, 0.0] + [0.0] * len(cross) + [0.0]
            out = ops.lincomb_rows(rows, [W_loss, W_kld])
            ind = [per[i][0][0].detach().sum() for i in range(M)]
        return {"loss": out[0], "reconstruction_loss": ind, "kld": out[1]}

    def modality_mixing(self, mods):
        return self.encode(mods)

    def forward(self, x, K=1):
        """mmvae_models.py:467-503: the container the evaluation code reads (shared / private posteriors, joint
        posterior, own / joint / cross reconstructions), missing modalities (`data` None, masks kept) included: the
        joint is the product of the PRESENT shared experts; a missing modality samples its shared code from the first
        present modality's posterior and its private code from N(0, I) (:489-493); cross reconstructions from a fresh
        shared draw of every other present modality (:499-502).  Noise in the reference's draw order."""
        if K != 1:
            raise NotImplementedError("dmvae.forward: K = 1 only on this path")
        missing, filled = self.get_missing_modalities(x)
        assert len(filled) > 0, "at least one modality must be present for forward call"
        D = self.n_latents
        enc_d = self.encode(x)
        mu_j, var_j = self.product_of_experts([enc_d[n]["shared"][0] for n in filled],
                                              [enc_d[n]["shared"][1] for n in filled])
        B, dev = mu_j.shape[0], mu_j.device
        draw = lambda d: self._draw(B, d, dev).unsqueeze(0)
        joint_d = normal(mu_j, var_j)
        z_joint = mu_j + var_j * draw(D)
        joint_dist, qz_xs, qz_private, zss, px_zs, joint_px_zs, cross_px_zs = {}, {}, {}, {}, {}, {}, {}
        for n, vae in self.vaes.items():
            present = n in filled
            joint_dist[n] = joint_d
            qz_xs[n] = normal(*enc_d[n]["shared"]) if present else None
            qz_private[n] = normal(*enc_d[n]["private"]) if present else None
            s_mu, s_lv = enc_d[n if present else filled[0]]["shared"]
            z_shared = s_mu + s_lv * draw(D)
            if present:
                p_mu, p_lv = enc_d[n]["private"]
                z_private = p_mu + p_lv * draw(vae.private_latents)
            else:
                z_private = draw(vae.private_latents)
            masks = x[n]["masks"]
            dec = lambda z: self._px(vae, *vae.dec({"latents": torch.cat([z, z_private], -1), "masks": masks}))
            zss[n] = {"latents": z_shared, "masks": masks}
            px_zs[n] = dec(z_shared)
            joint_px_zs[n] = dec(z_joint)
            cross_px_zs[n] = {}
            for m in filled:
                if m != n:
                    c_mu, c_lv = enc_d[m]["shared"]
                    cross_px_zs[n][m] = dec(c_mu + c_lv * draw(D))
        return self.make_output_dict(qz_xs, px_zs, zss, joint_dist, qz_private, None, joint_px_zs, cross_px_zs)
