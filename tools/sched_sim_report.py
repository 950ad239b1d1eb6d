import sys; sys.path.insert(0,'tools')
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    import sched_sim2 as S
for path in sys.argv[1:]:
    pixels = S.parse(path); n=len(pixels)
    b = S.sim_base(pixels[:64*24],24).report('base '+path)
    for NW,PJ,eso in ((8,4,False),(8,6,False),(8,8,False),(8,7,True),(8,8,True),(6,6,True)):
        px = [pixels[j % n] for j in range(6*(NW+PJ)*64)]
        k = S.sim_colpool(px, 6*(NW+PJ), NW=NW, PJ=PJ, swap_cost=35, vote_cost=60, empty_stack_only=eso).report(f'colpool NW={NW} PJ={PJ} swap35 vote60 empty_stack_only={eso}')
        print(f'     -> x{b/k:.2f}')
