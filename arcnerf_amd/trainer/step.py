"""One iteration as the reference's trainer runs it (arcnerf/trainer/arcnerf_trainer.py:494-548 train_epoch, :319-333 step_optimize)."""
import torch


def step_optimize(model, feed_in, loss_factory, optimizer, ema=None, epoch=0, total_epoch=300000, get_progress=False, clip_value=0.0):
    """output = model(feed_in); loss = loss_factory(feed_in, output); zero_grad; backward; [clip]; optimizer.step; ema.ema_step
    loss_factory: callable (feed_in, output) -> {'sum': tensor, ...} like arcnerf.loss.AllLoss (or a tensor)."""
    output = model(feed_in, get_progress=get_progress, cur_epoch=epoch, total_epoch=total_epoch)
    loss = loss_factory(feed_in, output)
    total = loss['sum'] if isinstance(loss, dict) else loss
    if not getattr(optimizer, 'zero_grad_on_step', False):     # (FusedAdam clears the gradients while it reads them)
        optimizer.zero_grad()
    total.backward()
    if clip_value > 0.0:
        torch.nn.utils.clip_grad_value_(model.parameters(), clip_value)
    optimizer.step()
    if ema is not None:
        ema.ema_step()
    return output, loss


def train_epoch(model, get_batch, loss_factory, optimizer, ema, pipeline, epoch, total_epoch=300000, stepper=None):
    """train_epoch's order: model.optimize(epoch) (the bound's periodic refresh), the dynamic batch size, then the step.
    get_batch(n_rays) -> feed_in dict; a callable with `wants_epoch = True` (trainer.TrainBatches: the Pipeline's crop / shuffle / batch
    fetch) is called as get_batch(n_rays, epoch).
    stepper: a trainer.FusedNgpStep (or any callable (feed_in, epoch) -> (output, loss)) in place of step_optimize.  A FusedNgpStep is also handed the batches of the next
    epochs (two by default) as long as neither model.optimize nor the batch-size rule can act at those epochs (then their "optimize,
    batch size, batch" commutes with this step): their marching runs on the second stream meanwhile."""
    draw = (lambda n, e: get_batch(n, e)) if getattr(get_batch, 'wants_epoch', False) else (lambda n, e: get_batch(n))
    if stepper is None:
        model.optimize(epoch)
        n_rays = pipeline.fetch_step_update_dynamic_bs(epoch, model)
        return step_optimize(model, draw(n_rays, epoch), loss_factory, optimizer, ema, epoch, total_epoch)
    feed_in = stepper.take_ahead(epoch) if hasattr(stepper, 'take_ahead') else None
    if feed_in is None:
        model.optimize(epoch)
        n_rays = pipeline.fetch_step_update_dynamic_bs(epoch, model)
        feed_in = draw(n_rays, epoch)
    if not hasattr(stepper, 'can_run_ahead'):
        return stepper(feed_in, epoch)
    # batches are drawn in epoch order, each exactly once, as far ahead as the stepper holds and the loop allows
    while stepper.ahead_room() > 0:
        e = stepper.next_ahead_epoch(epoch)
        if e >= total_epoch or not stepper.can_run_ahead(e) or pipeline.will_update_dynamic_bs(e):
            break
        stepper.hold_ahead(e, draw(pipeline.get_info('n_rays'), e))
    return stepper(feed_in, epoch, next_feed_in=stepper.ahead())
