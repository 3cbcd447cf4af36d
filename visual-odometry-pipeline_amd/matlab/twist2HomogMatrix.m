function H = twist2HomogMatrix(twist)
% Drop-in for utils/exe9_BA/twist2HomogMatrix.m: closed-form exp of the se(3) element [v; w].
H = vohip_mex('twist2homog', double(twist(:)));
end
