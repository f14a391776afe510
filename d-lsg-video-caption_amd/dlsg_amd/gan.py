"""the `DiscV2` critic and the WGAN-GP iteration `train_debug.py` really trains
(models/model.py:110-168, models/layer.py:661-715, run_gun.py:153-234,339-398).

Both sides of an iteration run on this repo's HIP kernels as hand-written launch schedules, no autograd inside:
  * generator -- both forwards of an iteration (the no-grad one of run_gun.py:167 and the trained one of :183), the ragged
    CrossEntropy, the whole backward and Adam through `Trainer` (engine.py); the critic's gradient w.r.t. the logits enters the
    backward as an extra d(logits) term (`Trainer.step(extra_dlogits=)`).  The reference detaches the proposals and attention
    weights before they reach the critic (run_gun.py:214-217), so d(logits) is the only path from the GAN loss into the generator.
  * critic -- `critic.CriticEngine`: forward, input backward, the derivative of both along the gradient penalty's direction, and the
    loss backward (the double backward of run_gun.py:362-371 as four explicit passes), then Adam on the critic's flat arena.
    Nothing of size (B, L, V) is materialised besides the generator's own logits: Conv1d(V -> 512, k = 1) is a linear map over the
    vocabulary axis, so real captions are a GATHER of its weight columns (run_gun.py:447-451 builds the one-hot), the
    gradient-penalty sample eps * real + (1 - eps) * fake is mixed AFTER the projection (the three critic forwards of a step share
    one logit projection and run as one 3B-caption batch), and |d mixed_logit / d mixed_captions|^2 = sum_l g_l (W W^T) g_l^T: a
    512 x 512 Gram matrix replaces the (B, L, V) gradient tensor.
All of it is exact up to fp32 reassociation; tests compare against the reference's own numbers (tests/golden/gan_*.npz).

`DiscV2.state_dict()` has the reference's keys and shapes (checkpoint key `model_d_state_dict`, run_gun.py:306).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .critic import CriticEngine, C as WIDTH, PW as PSL_WIDTH
from .model import _ArenaModule, _ALIGN, _capture_stream, _copy_h2d
from .modules import LatentPSL, SelfAttention


class _Residual(nn.Module):
    """parameter holder with the reference's key `res_block.1.{weight,bias}` (sublayer.py:107-115)"""

    def __init__(self, dim):
        super().__init__()
        self.res_block = nn.Sequential(nn.ReLU(True), nn.Conv1d(dim, dim, 3, padding=1))


class _PairScorer(nn.Module):
    """JointEmbedVideoModel2 parameters (sublayer.py:292-303)"""

    def __init__(self, h):
        super().__init__()
        self.classify = nn.Linear(h, 1)
        self.visual_embed = nn.Sequential(nn.Linear(h, h), nn.Tanh())
        self.sent_embed = nn.Sequential(nn.Linear(h, h), nn.Tanh())


class _ProposalScore(nn.Module):
    """PSLScore2 parameters (layer.py:661-688)"""

    def __init__(self, num_psl, num_top):
        super().__init__()
        self.psl_scorer = _PairScorer(WIDTH)
        self.psl_embed = nn.Sequential(nn.Linear(PSL_WIDTH, WIDTH), nn.Tanh(), nn.LayerNorm(WIDTH))
        self.psl_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(WIDTH), nn.Dropout(0.3))
        self.att_norm = nn.Sequential(nn.Linear(WIDTH, WIDTH), nn.Tanh(), nn.LayerNorm(WIDTH))
        self.num_top = num_top
        self.select = num_psl > num_top


class _CriticFn(torch.autograd.Function):
    """`DiscV2.forward` for torch autograd: the forward schedule, and a FIRST-order backward (inputs and parameters) -- what the
    generator step of run_gun.py:214-231 needs.  The gradient penalty's double backward is not offered through autograd: it is
    `CriticEngine.update_gradients` (GanTrainer.train_disc)."""

    @staticmethod
    def forward(ctx, D, inputs, obj, mot, smask, alpha, *params):
        eng, ops = D.engine, D.ops
        B, L, V = inputs.shape
        ws = eng.prepare(inputs.device, B, L, V, smask, 1)
        logits_tm = torch.empty(L, B, V, dtype=torch.float32, device=inputs.device)
        ops.permute_tb(inputs.contiguous().view(B, L, V), logits_tm)          # (B,L,V) -> (L,B,V): rows of one word are dense
        seed = eng.next_seed()
        eng.proposals(ws, obj.contiguous(), mot.contiguous(), alpha, smask)
        out = eng.score(ws, logits_tm, seed)
        ctx.D, ctx.ws, ctx.seed, ctx.logits_tm = D, ws, seed, logits_tm
        return out.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        D, ws = ctx.D, ctx.ws
        eng, ops = D.engine, D.ops
        B, L, V = ws.B, ws.L, ws.V
        dl_tm = torch.empty(L, B, V, dtype=torch.float32, device=d_out.device)
        eng.score_backward(ws, ctx.logits_tm, d_out.contiguous(), ctx.seed, dl_tm)
        d_in = torch.empty(B, L, V, dtype=torch.float32, device=d_out.device)
        ops.permute_tb(dl_tm, d_in)
        G = D.grad_views()
        grads = tuple(G[name].clone() if p.requires_grad else None for name, p in D.named_parameters())
        return (None, d_in, None, None, None, None) + grads


class DiscV2(_ArenaModule):
    """Critic of the visual GAN.  forward(inputs (B,L,V), obj (B,P,1024), mot (B,P,1024), att_mask (B,L,L), alpha_all (B,L,2P))
    -> (B,) scores, the reference's call (models/model.py:143)."""

    def __init__(self, opt, vocab_size):
        super().__init__()
        if opt.visual_hidden_size != PSL_WIDTH:
            raise ValueError('DiscV2 takes %d-wide proposals (models/layer.py:666); visual_hidden_size = %d'
                             % (PSL_WIDTH, opt.visual_hidden_size))
        self.dim = WIDTH
        self.num_top = opt.num_topk
        self.seq_len = opt.max_words
        self.num_psl = opt.num_proposals
        self.block = nn.Sequential(_Residual(WIDTH))
        self.conv1d = nn.Conv1d(vocab_size, WIDTH, 1)
        self.lstm = nn.LSTM(WIDTH, WIDTH, batch_first=True, bidirectional=False)       # parameters only: critic._forward runs it
        self.layer_norm = nn.LayerNorm(WIDTH)
        self.att = SelfAttention(WIDTH, WIDTH, WIDTH, 0.3)
        self.att_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(WIDTH))
        self.motion_psl_score = _ProposalScore(opt.num_proposals, self.num_top)
        self.obj_psl_score = _ProposalScore(opt.num_proposals, self.num_top)
        self.text_sum = LatentPSL(WIDTH, 1)
        self.fusion = nn.Parameter(torch.empty(2, WIDTH))
        nn.init.xavier_uniform_(self.fusion, gain=nn.init.calculate_gain('tanh'))
        self._engine = None

    def __getstate__(self):
        st = super().__getstate__()
        st['_engine'] = None
        return st

    @property
    def engine(self):
        if self._engine is None:
            self._engine = CriticEngine(self)
        return self._engine

    def forward(self, inputs, obj_proposals, motion_proposals, att_mask=None, alpha_all=None):
        if att_mask is None or alpha_all is None:
            raise ValueError('DiscV2 needs att_mask (B,L,L) and alpha_all (B,L,2P) (models/model.py:143-166 reads both)')
        # run_gun.py:164-166 builds att_mask as the outer product of the non-<pad> word mask; its first row is that mask
        smask = att_mask[:, 0, :].to(torch.float32).contiguous()
        self.flatten_parameters_()
        return _CriticFn.apply(self, inputs.to(torch.float32), obj_proposals.detach(), motion_proposals.detach(), smask,
                               alpha_all.detach().to(torch.float32), *self.parameters())


def attention_mask(captions):
    """run_gun.py:164-166: (B,L,L) outer product of the non-<pad> mask"""
    seq = (captions > 0).to(torch.float32)
    return seq.unsqueeze(2) * seq.unsqueeze(1)


class GANLambdaHandler(object):
    """utils/utils.py:196-265: weight of the generator's GAN loss.  Stable at `gan_lambda` while the running caption loss
    (last 200 steps) does not rise; when its newer half exceeds the older half by 4 % the weight follows one half-cosine dip
    down to 0.006 and back over 500 steps.  `cap_list` is what the reference stores in its checkpoints."""
    WIDTH, PERIOD, LOW = 200, 500, 0.006

    def __init__(self, total_step, gan_lambda, cap_list=None):
        self.cap_list = list(cap_list) if cap_list is not None else []
        self.total_step = total_step
        self.current_step = 0
        self.counter = self.PERIOD
        self.current_schedule_step = 0
        self.start_gan_lambda = gan_lambda
        self.low_gan_lambda = self.LOW
        amp = (gan_lambda - self.LOW) / 2
        k = np.arange(self.PERIOD)
        # sin(pi * x / PERIOD) on x in [PERIOD/2, 3 PERIOD/2) resp. [3 PERIOD/2, 5 PERIOD/2): start high -> low -> high
        self.decrease_schedule = (np.sin(np.pi * (k + self.PERIOD // 2) / self.PERIOD) * amp + amp + self.LOW).tolist()
        self.increase_schedule = (np.sin(np.pi * (k + 3 * self.PERIOD // 2) / self.PERIOD) * amp + amp + self.LOW).tolist()
        self.current_lambda = gan_lambda
        self.state = 0                      # 0 stable, 1 decrease, 2 increase (never entered by the reference either)

    def update_gan_lambda(self, epoch, i, cap_loss):
        self.current_step = i - 1 + epoch * self.total_step
        self.cap_list.append(cap_loss)
        if len(self.cap_list) <= self.WIDTH:
            return
        self.cap_list = self.cap_list[-self.WIDTH:]
        if self.state == 0:
            half = self.WIDTH // 2
            if np.mean(self.cap_list[half:]) > 1.04 * np.mean(self.cap_list[:half]):
                self.state = 1
        elif self.current_schedule_step == self.counter - 1:
            self.current_schedule_step = 0
            self.state = 0

    def get_current_lambda(self):
        if self.state != 0:
            sched = self.decrease_schedule if self.state == 1 else self.increase_schedule
            self.current_lambda = sched[self.current_schedule_step]
            self.current_schedule_step += 1
        return self.current_lambda


class GanTrainer(object):
    """One `RunGAN.train` iteration (run_gun.py:147-234) on the HIP kernels.

        it = GanTrainer(model, D); out = it.iteration(frames, regions, captions, cap_lens, tf_ratio, epoch, i)

    `trainer` is the generator's `dlsg_amd.Trainer`; on a GPU its step is two hipGraph replays (forward + CrossEntropy |
    backward + Adam) with the critic's term between them; a critic update (`CriticEngine.update_gradients` + Adam on the critic's
    arena) and the critic's term of the generator step are replayed hipGraphs of this repo's kernels as well
    (use_graphs=False: everything kernel by kernel)."""

    def __init__(self, model, D, lr=1.6e-4, betas=(0.5, 0.9), num_D=5, gan_lambda=0.01, total_step=1, cap_list=None,
                 process_group=None, world_size=1, use_graphs=None, eps=1e-8):
        self._loss_host = None            # pinned word the generator step's caption loss is copied into (iteration)
        from .model import Trainer
        self.model, self.D = model, D
        on_gpu = next(D.parameters()).is_cuda
        self.use_graphs = on_gpu if use_graphs is None else (use_graphs and on_gpu)
        # the generator's step replays forward + CrossEntropy and backward + Adam around the GAN term (Trainer.step)
        self.trainer = Trainer(model, lr=lr, betas=betas, process_group=process_group, world_size=world_size,
                               use_graphs=self.use_graphs)
        if D._ops_obj is None and model._ops_obj is not None:
            D.set_ops(model._ops_obj)
        # the critic's Adam (run_gun.py:100): moments over the critic's flat arena, one launch per update
        self.lr_D, self.betas_D, self.eps_D = lr, tuple(betas), eps
        self.t_D = 0
        self.m_D = self.v_D = None
        self.num_D = num_D
        self.lambda_handler = GANLambdaHandler(total_step, gan_lambda, cap_list)
        self.world_size, self.pg = world_size, process_group
        self.eps_source = None              # tests: callable(k) -> (B,1,1) tensor instead of torch.rand
        self._cg = {}

    # ------------------------------------------------------------------ the critic's optimizer (run_gun.py:100)
    def _bind_D(self):
        D = self.D
        D.flatten_parameters_()
        if self.m_D is None or self.m_D.shape != D._flat.shape or self.m_D.device != D._flat.device:
            self.m_D, self.v_D = torch.zeros_like(D._flat), torch.zeros_like(D._flat)
            self._cg.clear()

    def _hyper_D(self):
        b1, b2 = self.betas_D
        return [self.lr_D / (1.0 - b1 ** self.t_D), math.sqrt(1.0 - b2 ** self.t_D)]

    def _adam_D(self, hyper=None):
        D = self.D
        D.ops.adam(D._flat, D._gflat, self.m_D, self.v_D, self.lr_D, self.betas_D[0], self.betas_D[1], self.eps_D, max(self.t_D, 1),
                   1.0 / self.world_size, hyper=hyper)

    def optimizer_d_state_dict(self):
        """`optimizer_d_state_dict` of a checkpoint (run_gun.py:307) in the layout of torch.optim.Adam(D.parameters()).state_dict()"""
        D = self.D
        self._bind_D()
        state = {}
        names = [n for n, _ in D.named_parameters()]
        for i, (name, p) in enumerate(D.named_parameters()):
            if self.t_D == 0 or not p.requires_grad:
                continue
            o = D._offsets[name]
            state[i] = {'step': torch.tensor(float(self.t_D)),
                        'exp_avg': self.m_D[o:o + p.numel()].view(p.shape).clone(),
                        'exp_avg_sq': self.v_D[o:o + p.numel()].view(p.shape).clone()}
        group = {'lr': self.lr_D, 'betas': tuple(self.betas_D), 'eps': self.eps_D, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(len(names)))}
        return {'state': state, 'param_groups': [group]}

    def load_optimizer_d_state_dict(self, sd):
        D = self.D
        self._bind_D()
        self.m_D.zero_(); self.v_D.zero_()
        steps = set()
        for i, (name, p) in enumerate(D.named_parameters()):
            st = sd['state'].get(i, sd['state'].get(str(i)))
            if st is None:
                continue
            o = D._offsets[name]
            self.m_D[o:o + p.numel()].view(p.shape).copy_(st['exp_avg'])
            self.v_D[o:o + p.numel()].view(p.shape).copy_(st['exp_avg_sq'])
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('per-parameter step counts differ (%s): not a state this trainer can resume' % sorted(steps))
        self.t_D = steps.pop() if steps else 0
        g = sd['param_groups'][0]
        self.lr_D, self.betas_D, self.eps_D = g['lr'], tuple(g['betas']), g['eps']
        self.reset_graphs()

    def critic_adam_step(self, grads):
        """one Adam step of the critic's optimizer on given gradients {parameter name: tensor}"""
        self._bind_D()
        G = self.D.grad_views()
        self.D._gflat.zero_()
        for n, t in grads.items():
            G[n].copy_(t)
        self.t_D += 1
        self._adam_D(hyper=torch.tensor(self._hyper_D(), dtype=torch.float32).to(self.D._flat.device))

    def reset_graphs(self):
        """Drop the captured critic graphs (a changed arena / optimizer state layout invalidates their pointers)."""
        self._cg.clear()

    def _rank_mean(self, value):
        """run_gun.py:433-437 `reduce_tensor`: the mean over ranks of a logged scalar.  Every rank then sees the same caption
        loss, so GANLambdaHandler switches its schedule at the same step everywhere (run_gun.py:202-203,212)."""
        if self.world_size <= 1:
            return float(value)
        import torch.distributed as dist
        t = value.detach().reshape(1).to(torch.float32).clone() if torch.is_tensor(value) else \
            torch.tensor([float(value)], dtype=torch.float32, device=next(self.D.parameters()).device)
        dist.all_reduce(t, group=self.pg)
        return float(t) / self.world_size

    def _allreduce_D(self):
        """sum of the critic's gradients over the ranks (DDP of `model_d`, run_gun.py:71-72; the mean's 1 / world is folded into
        the Adam launch): ONE collective on the flat gradient arena"""
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(self.D._gflat, group=self.pg)

    # ------------------------------------------------------------------ critic updates (run_gun.py:339-381)
    @staticmethod
    def _sig(ts):
        return tuple((t.data_ptr(), tuple(t.shape), tuple(t.stride())) for t in ts)

    def _critic_static(self, captions, logits_tm, obj, mot, smask, alpha, alias):
        """static buffers + captured graphs of one batch shape: graph P = proposals, graph A = update_gradients, graph B = Adam (the
        gradient all-reduce of a multi-GPU run sits between A and B).  alias: the generator's outputs (logits, proposals, attention
        weights) ARE static buffers of the generator's own captured step (Trainer.forward_only): the critic's graphs read them in
        place -- no staging copies (6.6 MB of logits per iteration at batch 64) -- and are keyed by their addresses."""
        D, eng = self.D, self.D.engine
        L, B, V = logits_tm.shape
        big = (logits_tm, obj, mot, alpha)
        key = ('D', B, L, V, tuple(obj.shape), D.training, self.lr_D, self._sig(big) if alias else None)
        st = self._cg.get(key)
        if st is not None:
            return st
        dev = logits_tm.device
        st = dict(captions=captions.clone(), smask=smask.clone(), eps=torch.zeros(B, device=dev),
                  seed=torch.zeros(1, dtype=torch.int64, device=dev), hyper=torch.zeros(2, device=dev), graphs=None, alias=alias)
        if alias:
            st.update(logits=logits_tm, obj=obj, mot=mot, alpha=alpha)
        else:
            st.update(logits=logits_tm.clone(), obj=obj.clone(), mot=mot.clone(), alpha=alpha.contiguous().clone())
        st['ws'] = eng.prepare(dev, B, L, V, st['smask'], 4)
        self._cg[key] = st
        return st

    def _capture_critic(self, st):
        eng = self.D.engine
        side = _capture_stream(st['logits'].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            eng.proposals(st['ws'], st['obj'], st['mot'], st['alpha'], st['smask'])       # warm-up on the capture stream
            eng.update_gradients(st['ws'], st['captions'], st['logits'], st['eps'], st['seed'])
            side.synchronize()
            pool = torch.cuda.graph_pool_handle()
            gP, gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(gP, pool=pool, stream=side, capture_error_mode='thread_local'):
                eng.proposals(st['ws'], st['obj'], st['mot'], st['alpha'], st['smask'])
            with torch.cuda.graph(gA, pool=pool, stream=side, capture_error_mode='thread_local'):
                st['stats'] = eng.update_gradients(st['ws'], st['captions'], st['logits'], st['eps'], st['seed'])
            with torch.cuda.graph(gB, pool=pool, stream=side, capture_error_mode='thread_local'):
                self._adam_D(hyper=st['hyper'])
        torch.cuda.current_stream().wait_stream(side)
        st['graphs'] = (gP, gA, gB)

    def train_disc(self, captions, logits_tm, obj, mot, smask, alpha, alias=False, as_tensor=False):
        """run_gun.py:339-381: num_D critic updates.  logits_tm (L,B,V): the generator's logits, time-major as its decoder writes
        them.  alias: see _critic_static.  Returns (mean loss_D, mean Wasserstein estimate) as floats -- or, as_tensor, as a
        2-element device tensor (means over ranks included) WITHOUT a host synchronisation: `iteration` reads it back together
        with the generator's losses, so the device does not drain between the critic updates and the generator step."""
        D, eng = self.D, self.D.engine
        self._bind_D()
        L, B, V = logits_tm.shape
        dev = logits_tm.device
        acc = torch.zeros(8, device=dev)
        if self.use_graphs:
            st = self._critic_static(captions, logits_tm, obj, mot, smask, alpha, alias)
            for k_, src in (('captions', captions), ('smask', smask)) + \
                    (() if alias else (('logits', logits_tm), ('obj', obj), ('mot', mot), ('alpha', alpha))):
                st[k_].copy_(src, non_blocking=True)
            if st['graphs'] is None:
                self._capture_critic(st)
            st['graphs'][0].replay()
        else:
            ws = eng.prepare(dev, B, L, V, smask, 4)
            eng.proposals(ws, obj.contiguous(), mot.contiguous(), alpha, smask)
        for k in range(self.num_D):
            eps = self.eps_source(k) if self.eps_source is not None else torch.rand(B, device=dev)
            eps = eps.reshape(B).to(device=dev, dtype=torch.float32)
            seed = eng.next_seed()
            self.t_D += 1
            if self.use_graphs:
                st['eps'].copy_(eps, non_blocking=True)
                _copy_h2d(st['seed'], [seed])
                _copy_h2d(st['hyper'], self._hyper_D())
                st['graphs'][1].replay()
                self._allreduce_D()
                st['graphs'][2].replay()
                acc += st['stats']
            else:
                stats = eng.update_gradients(ws, captions, logits_tm, eps, seed)
                self._allreduce_D()
                # (the bias corrections as the replayed launch reads them: a device word, so both forms give the same bits)
                self._adam_D(hyper=torch.tensor(self._hyper_D(), dtype=torch.float32).to(dev))
                acc += stats
        acc /= self.num_D
        if as_tensor:
            t = torch.stack([acc[0], acc[4]])
            if self.world_size > 1:
                import torch.distributed as dist
                dist.all_reduce(t, group=self.pg)
                t /= self.world_size
            return t
        return self._rank_mean(acc[0]), self._rank_mean(acc[4])

    # ------------------------------------------------------------------ the critic's term of the generator step (run_gun.py:214-231)
    def _generator_term(self, logits_tm, obj, mot, smask, alpha, out):
        """loss_G = -D(tokens).mean() (run_gun.py:214-217): enqueues the critic's forward and its backward down to the 512-wide
        projection, returns finish(scale) -> device scalar of loss_G, which launches the last product:
        out (L,B,V) = scale * d loss_G / d logits (gan_lambda is a host number that changes every step)."""
        D, eng = self.D, self.D.engine
        L, B, V = logits_tm.shape
        dev = logits_tm.device
        if not self.use_graphs:
            ws = eng.prepare(dev, B, L, V, smask, 1)
            eng.proposals(ws, obj.contiguous(), mot.contiguous(), alpha, smask)
            score = eng.score(ws, logits_tm, eng.next_seed())
            dhf = eng.score_backward(ws, logits_tm, ws.d_outG, eng.seed_last, None, params=False)

            def finish(scale):
                eng.dlogits(ws, dhf, out, scale)
                return -score.mean()
            return finish
        # (inside Trainer.step's cut graphs the logits, proposals and attention weights are static buffers: read in place)
        key = ('G', B, L, V, tuple(obj.shape), D.training, self._sig((logits_tm, obj, mot, alpha)))
        st = self._cg.get(key)
        if st is None:
            st = dict(logits=logits_tm, obj=obj, mot=mot, smask=smask.clone(), alpha=alpha,
                      seed=torch.zeros(1, dtype=torch.int64, device=dev), graph=None)
            st['ws'] = eng.prepare(dev, B, L, V, st['smask'], 1)
            self._cg[key] = st
        st['smask'].copy_(smask, non_blocking=True)
        _copy_h2d(st['seed'], [eng.next_seed()])
        if st['graph'] is None:
            side = _capture_stream(dev)
            side.wait_stream(torch.cuda.current_stream())

            def run():
                eng.proposals(st['ws'], st['obj'], st['mot'], st['alpha'], st['smask'])
                st['score'] = eng.score(st['ws'], st['logits'], st['seed'])
                st['dhf'] = eng.score_backward(st['ws'], st['logits'], st['ws'].d_outG, st['seed'], None, params=False)
            with torch.cuda.stream(side):
                run()
                side.synchronize()
                st['graph'] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(st['graph'], stream=side, capture_error_mode='thread_local'):
                    run()
            torch.cuda.current_stream().wait_stream(side)
        st['graph'].replay()

        def finish(scale):
            eng.dlogits(st['ws'], st['dhf'], out, scale)
            return -st['score'].mean()
        return finish

    def iteration(self, frames, regions, captions, cap_lens, tf_ratio, epoch=0, i=1, max_len=26):
        model, D = self.model, self.D
        captions = captions[:, :max_len].contiguous()
        smask = (captions > 0).to(torch.float32)                                          # run_gun.py:164-166 (its outer product)
        # ---- Train D: the generator's outputs are constants here (run_gun.py:167-174)
        fwd = self.trainer.forward_only(frames, regions, captions, tf_ratio, max_len, time_major=True)   # replayed once the step is captured
        if fwd is None:
            with torch.no_grad():
                f_caption, obj, mot, alpha = model(frames, regions, captions, max_len, tf_ratio)
            logits_tm = torch.empty(f_caption.shape[1], f_caption.shape[0], f_caption.shape[2], dtype=torch.float32, device=f_caption.device)
            D.ops.permute_tb(f_caption.contiguous(), logits_tm)                           # (B,L,V) -> (L,B,V)
        else:
            logits_tm, obj, mot, alpha = fwd
        d_stats = self.train_disc(captions, logits_tm, obj, mot, smask, alpha, alias=fwd is not None, as_tensor=True)
        # ---- Train the captioning model (run_gun.py:180-234)
        out = {}

        def gan_term(logits_tm, sv, dst=None):
            """d(gan_lambda * loss_G) / d logits, time-major (L,B,V); called between the HIP forward and backward"""
            s = sv['dec']
            obj_, mot_ = sv['dec_gsrc'][0].detach(), sv['dec_gsrc'][1].detach()
            alpha_ = s['ALPHA'].transpose(0, 1).detach()
            g = dst if dst is not None else torch.empty_like(logits_tm)
            # the critic's forward and backward do not depend on gan_lambda: they are enqueued BEFORE the caption loss is read back
            # (a host synchronisation), so the device works through them while the host updates the weight
            # the caption loss goes to the host through a copy + event issued BEFORE the critic's launches: the host has it (and
            # with it gan_lambda) while the device is still working through them, and enqueues the rest of the step without a gap
            pending = None
            if self.world_size <= 1 and sv['loss_dev'].is_cuda:
                if self._loss_host is None:
                    self._loss_host = torch.empty(1, dtype=torch.float32).pin_memory()
                self._loss_host.copy_(sv['loss_dev'].detach().reshape(1), non_blocking=True)
                pending = torch.cuda.Event()
                pending.record()
            finish = self._generator_term(logits_tm.detach(), obj_, mot_, smask, alpha_, g)
            out['cap_loss_dev'] = sv['loss_dev']
            # the reference updates lambda from the caption loss of THIS step before using it (run_gun.py:210,224)
            # -- with several ranks the all-reduced mean, as the reference feeds it (run_gun.py:202-203,212)
            if pending is not None:
                pending.synchronize()
                out['cap_loss_record'] = float(self._loss_host[0])
            else:
                out['cap_loss_record'] = self._rank_mean(sv['loss_dev'])
            self.lambda_handler.update_gan_lambda(epoch, i, out['cap_loss_record'])
            out['gan_lambda'] = self.lambda_handler.get_current_lambda()
            out['loss_G'] = finish(out['gan_lambda'])
            return g
        gan_term.takes_dst = True
        cap_loss = self.trainer.step(frames, regions, captions, cap_lens, tf_ratio, max_len=max_len, extra_dlogits=gan_term)
        # cap_loss / loss_G: this rank's values (what its backward used); *_record: the means over ranks the reference logs
        # ONE read-back for the iteration's remaining scalars (the caption loss was read once already, for gan_lambda)
        loss_G_record = self._rank_mean(out['loss_G']) if self.world_size > 1 else None
        word = getattr(model.ops, 'persist_word_or_none', lambda: None)()         # the persistent kernels' time-out word rides along
        vals = torch.cat([cap_loss.detach().reshape(1).float(), out['loss_G'].detach().reshape(1).float(), d_stats.float()] +
                         ([word.float()] if word is not None else [])).tolist()
        out.update(cap_loss=vals[0], loss_G_record=vals[1] if loss_G_record is None else loss_G_record, loss_G=vals[1],
                   loss_D=vals[2], wasserstein=vals[3])
        out['total_loss'] = out['cap_loss'] + out['loss_G'] * out['gan_lambda']
        out.pop('cap_loss_dev')
        check = getattr(model.ops, 'check_persistent', None)
        if check is not None:
            code = int(vals[4]) if word is not None else 0
            if self.world_size > 1:
                # (every rank takes part, whether or not one of its own kernels ever created the word)
                from .comm import _agree_min
                code = _agree_min(code, self.trainer.pg, negate=True)      # every rank raises, or none
            try:
                check(code=code)           # read back with the losses: the device was idle, the time-out word is final
            except RuntimeError:
                # graphs captured on the persistent schedule must not be replayed again: the generator's are captured anew on
                # the step-by-step BiLSTM; the critic's LSTM has no such form (the message says so)
                self._cg.clear()
                self.trainer._graphs = None
                raise
        return out


# ------------------------------------------------------------------ checkpoints (run_gun.py:302-310, 53-61, 92-109)
def save_checkpoint(path, epoch, gan):
    """The dict RunGAN stores: epoch, model_state_dict, optimizer_state_dict (torch.optim.Adam layout), model_d_state_dict,
    optimizer_d_state_dict, cap_list."""
    torch.save({'epoch': epoch,
                'model_state_dict': {k: v.detach().cpu() for k, v in gan.model.state_dict().items()},
                'optimizer_state_dict': gan.trainer.optimizer_state_dict(),
                'model_d_state_dict': {k: v.detach().cpu() for k, v in gan.D.state_dict().items()},
                'optimizer_d_state_dict': gan.optimizer_d_state_dict(),
                'cap_list': np.array(gan.lambda_handler.cap_list)}, path)


def load_checkpoint(path, gan, map_location=None):
    """Resume a GanTrainer from a checkpoint written by save_checkpoint or by the reference trainer.  Returns the epoch."""
    ck = torch.load(path, map_location=map_location or 'cpu', weights_only=False)
    gan.model.load_state_dict(ck['model_state_dict'])
    gan.trainer.load_optimizer_state_dict(ck['optimizer_state_dict'])
    gan.D.load_state_dict(ck['model_d_state_dict'])
    gan.load_optimizer_d_state_dict(ck['optimizer_d_state_dict'])
    h = gan.lambda_handler
    gan.lambda_handler = GANLambdaHandler(h.total_step, h.start_gan_lambda, cap_list=ck['cap_list'])
    return ck['epoch']
